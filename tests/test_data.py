"""TFRecord input path (reference train.py:71-103, inference.py:67-96) without TensorFlow: framing, tf.Example wire
format, and the device decode against the NumPy restatement of _parse_image_function (bit-exact: casts, crops and one
power-of-two scale)."""
import os

import numpy as np
import pytest
import torch

from strajnet_amd import data as D

# tf.train.Example(features=Features(feature={'a': Feature(bytes_list=BytesList(value=[b'xyz']))})).SerializeToString(),
# written out by hand from the protobuf wire format (field 1 length-delimited at every level)
EXAMPLE_A_XYZ = bytes.fromhex('0a0e0a0c0a016112070a050a0378797a')


def _raw_example(rng, grid, out, test=False):
    ex = {}
    for name, (dt, shape, crop, scale) in D.feature_spec(grid, out, test).items():
        n = int(np.prod(shape))
        if dt == 'bool':
            a = (rng.random(n) < 0.3).astype(np.bool_)
        elif dt == 'int8':
            a = rng.integers(-128, 128, n).astype(np.int8)
        elif dt == 'float32':
            a = rng.normal(size=n).astype(np.float32)
        else:
            a = rng.normal(size=n).astype(np.float64) * 40
        ex[name] = a.tobytes()
    return ex


def test_crc32c_and_example_wire_format():
    assert D.crc32c(b'123456789') == 0xE3069283                      # CRC-32C check value (RFC 3720 B.4)
    assert D.crc32c(b'') == 0
    assert D.serialize_example({'a': b'xyz'}) == EXAMPLE_A_XYZ
    assert {k: bytes(v) for k, v in D.parse_example(EXAMPLE_A_XYZ).items()} == {'a': b'xyz'}
    big = {'ogm': bytes(range(256)) * 40, 'scenario/id': b'abc-123', 'empty': b''}
    assert {k: bytes(v) for k, v in D.parse_example(D.serialize_example(big)).items()} == big


def test_tfrecord_roundtrip_and_corruption(tmp_path):
    p = os.path.join(tmp_path, 'r.tfrecords')
    payloads = [D.serialize_example({'a': bytes([i]) * (i * 100 + 1)}) for i in range(4)]
    D.write_tfrecord(p, payloads)
    assert list(D.read_tfrecord(p, check_data_crc=True)) == payloads
    # masked CRC of the 8-byte length header of a 16-byte record, as TF writes it
    head = open(p, 'rb').read(12)
    assert int.from_bytes(head[:8], 'little') == len(payloads[0]) and int.from_bytes(head[8:], 'little') == D.masked_crc(head[:8])
    raw = bytearray(open(p, 'rb').read())
    raw[14] ^= 0xFF                                                   # flip a payload byte of record 0
    open(p, 'wb').write(raw)
    with pytest.raises(ValueError):
        list(D.read_tfrecord(p, check_data_crc=True))
    raw[14] ^= 0xFF
    raw[3] ^= 0x01                                                    # corrupt the length
    open(p, 'wb').write(raw)
    with pytest.raises(ValueError):
        list(D.read_tfrecord(p))


def test_oracle_parse_shapes():
    from oracle import np_ref
    rng = np.random.default_rng(0)
    ex = _raw_example(rng, 64, 32)
    r = np_ref.parse_image_function(ex, 64, 32)
    assert r['ogm'].shape == (64, 64, 11, 2) and set(np.unique(r['ogm'])) <= {0.0, 1.0}
    assert r['gt_flow'].shape == (8, 32, 32, 2) and r['gt_obs_ogm'].shape == (8, 32, 32, 1)
    assert np.abs(r['map_image']).max() <= 0.5 and r['actors'].dtype == np.float32
    full = np.frombuffer(ex['gt_flow'], np.float32).reshape(8, 64, 64, 2)
    assert np.array_equal(r['gt_flow'], full[:, 16:48, 16:48])


@pytest.mark.gpu
@pytest.mark.parametrize('grid,out,test', [(64, 32, False), (64, 32, True), (512, 256, False)])
def test_decode_batch_matches_oracle(lib_built, tmp_path, grid, out, test):
    """records -> file -> reader -> parser -> device decode == np_ref.parse_image_function, bit for bit; (512, 256) is the
    reference's real record geometry (one 36 MB example)."""
    from oracle import np_ref
    rng = np.random.default_rng(1)
    B = 3 if grid == 64 else 1
    exs = [_raw_example(rng, grid, out, test) for _ in range(B)]
    if test:
        for i, e in enumerate(exs):
            e['scenario/id'] = f'scn{i}'.encode()
    p = os.path.join(tmp_path, 'd.tfrecords')
    D.write_tfrecord(p, [D.serialize_example(e) for e in exs])
    got = None
    for batch in D.batches(D.read_tfrecord(p), B):
        got = D.decode_batch(batch, 'cuda', grid, out, test)
    refs = [np_ref.parse_image_function(e, grid, out, test) for e in exs]
    for name in refs[0]:
        want = np.stack([r[name] for r in refs], 0)
        assert tuple(got[name].shape) == want.shape, name
        assert np.array_equal(got[name].cpu().numpy(), want), name
    if test:
        assert got['scenario/id'] == [b'scn0', b'scn1', b'scn2']
    with pytest.raises(ValueError):
        bad = dict(exs[0]); bad['actors'] = bad['actors'][:-8]
        D.decode_batch([bad], 'cuda', grid, out, test)


@pytest.mark.gpu
def test_host_feed_lands_every_batch(lib_built):
    """data.HostFeed: three consecutive batches written into the pinned host buffers arrive bit-exactly in the static inputs (float32
    tensors as they are, raw bool / int8 bytes expanded like _parse_image_function does), each one landing while the next is uploaded."""
    import torch
    from strajnet_amd.data import HostFeed
    g = torch.Generator().manual_seed(0)
    static = {'ogm': torch.zeros((2, 64, 64, 11, 2), device='cuda'), 'map_img': torch.zeros((2, 64, 64, 3), device='cuda'),
              'flow': torch.zeros((2, 64, 64, 2), device='cuda'), 'big': torch.zeros((3, 1 << 20), device='cuda')}       # 12 MB: several pieces
    host = {'ogm': torch.zeros((2, 64, 64, 11, 2), dtype=torch.uint8).pin_memory(), 'map_img': torch.zeros((2, 64, 64, 3), dtype=torch.uint8).pin_memory(),
            'flow': torch.zeros((2, 64, 64, 2)).pin_memory(), 'big': torch.zeros((3, 1 << 20)).pin_memory(), 'ignored': torch.zeros(4).pin_memory()}
    feed = HostFeed(static, host, raw={'ogm': 'bool', 'map_img': 'int8'})

    def fill(seed):
        g.manual_seed(seed)
        host['ogm'].copy_((torch.rand(host['ogm'].shape, generator=g) < 0.3).to(torch.uint8) * 7)           # any non-zero byte is True
        host['map_img'].copy_(torch.randint(-128, 128, host['map_img'].shape, generator=g, dtype=torch.int16).to(torch.int8).view(torch.uint8))
        host['flow'].copy_(torch.randn(host['flow'].shape, generator=g))
        host['big'].copy_(torch.randn(host['big'].shape, generator=g))
        return {'ogm': (host['ogm'] != 0).float(), 'map_img': host['map_img'].view(torch.int8).float() / 256.0, 'flow': host['flow'].clone(), 'big': host['big'].clone()}
    want = fill(1)
    feed.start()
    for step in range(3):
        feed.wait_uploaded()              # the batch in flight has left the host buffers: refill them with the next one
        nxt = fill(2 + step)
        feed.land()                       # lands `want`, starts uploading `nxt`
        torch.cuda.synchronize()
        for k in want:
            assert torch.equal(static[k].cpu(), want[k]), (step, k)
        want = nxt
    feed.wait_uploaded()
    feed.close()
