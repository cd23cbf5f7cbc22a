"""TF-checkpoint bundle reader / writer and the reference object-graph name map (strajnet_amd/checkpoint.py) -- host logic.

FORMAT UNPINNED against TensorFlow itself (none here, and the reference repository ships no checkpoint): the byte-level tests
below are assembled by hand from the published table / bundle layout, the rest pins reader and writer against each other."""
import os
import struct

import numpy as np
import pytest

from strajnet_amd import checkpoint as ck
from strajnet_amd.data import crc32c, _crc32c_py

CFG = dict(input_size=(256, 256), window_size=8, embed_dim=96, depths=[2, 2, 2], num_heads=[3, 6, 12])


def test_crc32c_known_answers_and_host_routine():
    assert _crc32c_py(b'123456789') == 0xE3069283                      # the CRC-32C check value
    assert _crc32c_py(bytes(32)) == 0x8A9136AA                          # RFC 3720 B.4: 32 zero bytes
    assert _crc32c_py(b'\xff' * 32) == 0x62A8AB43                       # RFC 3720 B.4: 32 0xFF bytes
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 8, 9, 63, 4097):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert crc32c(b) == _crc32c_py(b)
        assert crc32c(b[3:], crc32c(b[:3])) == _crc32c_py(b)            # running CRC
    a = rng.standard_normal((5, 7)).astype(np.float32)
    assert crc32c(a) == _crc32c_py(a.tobytes())
    assert ck._unmask(ck._mask(0x12345678)) == 0x12345678


def test_block_layout_by_hand():
    """Two keys sharing a prefix inside one restart interval: entry = varint shared, non-shared, value length, key tail,
    value; then the restart offsets (u32) and their count."""
    body = ck._build_block([(b'ab', b'X'), (b'abc', b'YZ')])
    assert body == bytes([0, 2, 1]) + b'abX' + bytes([2, 1, 2]) + b'cYZ' + struct.pack('<II', 0, 1)
    assert list(ck._block_entries(body)) == [(b'ab', b'X'), (b'abc', b'YZ')]
    assert ck._build_block([]) == struct.pack('<II', 0, 1)              # empty block: one restart at 0


def test_table_footer_and_magic_by_hand():
    t = ck.write_table([(b'k', b'v')])
    assert t[-8:] == bytes.fromhex('57fb808b247547db') and len(t) >= 48
    data = ck._build_block([(b'k', b'v')])
    assert t[:len(data)] == data and t[len(data)] == 0                  # block, then compression type 0 ...
    assert struct.unpack('<I', t[len(data) + 1:len(data) + 5])[0] == ck._mask(_crc32c_py(data + b'\0'))   # ... and its CRC
    foot = t[-48:]
    moff, msize, i = ck._handle(foot, 0)
    ioff, isize, j = ck._handle(foot, i)
    assert moff == len(data) + 5 and msize == 8 and ioff == moff + 13
    assert foot[j:40] == bytes(40 - j)
    assert ck.read_table(t) == [(b'k', b'v')]


def test_table_roundtrip_many_blocks_and_corruption():
    rng = np.random.default_rng(1)
    items = sorted({(b'layer%04d/w%d' % (rng.integers(0, 3000), rng.integers(0, 9))): rng.bytes(int(rng.integers(0, 40)))
                    for _ in range(2000)}.items())
    for bs in (64, 1024, 1 << 20):
        t = ck.write_table(items, block_size=bs)
        assert ck.read_table(t) == items
    t = bytearray(ck.write_table(items, block_size=1024))
    t[100] ^= 1
    with pytest.raises(ValueError, match='checksum'):
        ck.read_table(bytes(t))
    with pytest.raises(ValueError, match='magic'):
        ck.read_table(b'\0' * 64)
    with pytest.raises(ValueError, match='increasing'):
        ck.write_table([(b'b', b''), (b'a', b'')])


def test_snappy_decoder_by_hand():
    assert ck.snappy_decompress(bytes([5, 4 << 2]) + b'hello') == b'hello'                      # literal, length 5
    # 'abcabcabcab': literal 'abc' then a 1-byte-offset copy of 8 bytes from 3 back (overlapping)
    assert ck.snappy_decompress(bytes([11, 2 << 2]) + b'abc' + bytes([((8 - 4) << 2) | 1, 3])) == b'abcabcabcab'
    # 2-byte-offset copy, and a 61-byte literal with the length in one extra byte
    lit = bytes(range(61))
    assert ck.snappy_decompress(bytes([63, 60 << 2, 60]) + lit + bytes([((2 - 1) << 2) | 2, 61, 0])) == lit + lit[:2]
    with pytest.raises(ValueError):
        ck.snappy_decompress(bytes([4, ((4 - 4) << 2) | 1, 9]))


def test_bundle_entry_by_hand():
    """BundleEntryProto {dtype = DT_FLOAT(1), shape {dim {size: 2} dim {size: 3}}, offset = 24, size = 24, crc32c}."""
    raw = bytes([0x08, 1, 0x12, 8, 0x12, 2, 0x08, 2, 0x12, 2, 0x08, 3, 0x20, 24, 0x28, 24, 0x35]) + struct.pack('<I', 0xDEADBEEF)
    assert ck._entry_bytes(1, (2, 3), 24, 24, 0xDEADBEEF) == raw
    e = ck._parse_entry(raw)
    assert (e['dtype'], e['shape'], e['offset'], e['size'], e['crc32c'], e['shard_id']) == (1, [2, 3], 24, 24, 0xDEADBEEF, 0)


def test_bundle_roundtrip_dtypes_and_checksum(tmp_path):
    p = str(tmp_path / 'sub' / 'ckpt.tf')
    rng = np.random.default_rng(2)
    tens = {'a/f32': rng.standard_normal((3, 4, 5)).astype(np.float32), 'a/f64': rng.standard_normal(7),
            'b/i64': rng.integers(-9, 9, (64, 64)), 'b/flag': rng.integers(0, 2, (4, 4)).astype(bool),
            'scalar': np.float32(3.5), 'empty': np.zeros((0, 3), np.float32)}
    w = ck.BundleWriter(p)
    for k, v in tens.items():
        w.add(k, v)
    w.add_string('note', b'hello bundle')
    w.finish()
    assert sorted(os.listdir(tmp_path / 'sub')) == ['ckpt.tf.data-00000-of-00001', 'ckpt.tf.index']
    r = ck.BundleReader(p)
    assert sorted(r.keys()) == sorted(list(tens) + ['note'])
    for k, v in tens.items():
        got = r.get(k)
        assert got.dtype == np.asarray(v).dtype and got.shape == np.shape(v) and np.array_equal(got, v)
    assert r.get('note') == b'hello bundle'
    with open(p + '.data-00000-of-00001', 'r+b') as f:                   # flip one bit of the first tensor
        f.seek(5); b = f.read(1); f.seek(5); f.write(bytes([b[0] ^ 4]))
    with pytest.raises(ValueError, match='checksum'):
        ck.BundleReader(p).get('a/f32')
    assert ck.BundleReader(p, verify=False).get('a/f32').shape == (3, 4, 5)


def test_string_entry_checksum_follows_tensor_bundle(tmp_path):
    """DT_STRING entries (the object graph): TF's WriteStringTensor extends the running CRC-32C with the element length as a
    uint32 (uint64 only above 2^32-1), then the 4 bytes of the masked length checksum, then the bytes.  The expected values are
    assembled here byte by byte, independently of BundleWriter.add_string; the reader verifies both checksums."""
    import struct
    p = str(tmp_path / 'c.tf')
    val = b'object graph bytes'
    w = ck.BundleWriter(p)
    w.add_string('s', val)
    w.finish()
    raw = open(p + '.data-00000-of-00001', 'rb').read()
    c_len = ck.crc32c(struct.pack('<I', len(val)))                         # 4-byte length word, NOT 8
    lcs = struct.pack('<I', ck._mask(c_len))
    assert raw == bytes([len(val)]) + lcs + val
    want = ck._mask(ck.crc32c(val, ck.crc32c(lcs, c_len)))
    r = ck.BundleReader(p)
    assert r.entry('s')['crc32c'] == want
    assert r.entry('s')['crc32c'] != ck._mask(ck.crc32c(val, ck.crc32c(lcs, ck.crc32c(struct.pack('<Q', len(val))))))
    assert r.get('s') == val
    with open(p + '.data-00000-of-00001', 'r+b') as f:                     # corrupt one string byte: the reader must notice
        f.seek(7); b = f.read(1); f.seek(7); f.write(bytes([b[0] ^ 1]))
    with pytest.raises(ValueError, match='checksum'):
        ck.BundleReader(p).get('s')


def test_object_graph_walk_prefers_graph_over_key_names(tmp_path):
    """The graph maps attribute paths to whatever checkpoint key the writer chose (Keras may pick layer_with_weights-N)."""
    def node(children, attr=None):
        body = b''
        for name, nid in children:
            body += ck._ld(1, ck._vint(1, nid) + ck._ld(2, name.encode()))
        if attr:
            body += ck._ld(2, ck._ld(1, b'VARIABLE_VALUE') + ck._ld(2, b'dense/kernel') + ck._ld(3, attr.encode()))
        return ck._ld(1, body)
    key = 'layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE'
    graph = node([('encoder', 1), ('layer_with_weights-0', 2)]) + node([('blocks', 3)]) + node([('kernel', 4)]) + \
        node([('0', 2)]) + node([], key)
    p = str(tmp_path / 'g')
    w = ck.BundleWriter(p)
    w.add(key, np.arange(6, dtype=np.float32).reshape(2, 3))
    w.add('plain/bias' + ck.VAR_SUFFIX, np.ones(3, np.float32))
    w.add_string(ck.OBJECT_GRAPH_KEY, graph)
    w.finish()
    r = ck.BundleReader(p)
    g = r.object_graph()
    assert len(g) == 5 and g[0]['children'] == {'encoder': 1, 'layer_with_weights-0': 2}
    assert g[4]['attributes'] == [('VARIABLE_VALUE', 'dense/kernel', key)]
    assert r.resolve('encoder/blocks/0/kernel') == key                   # through a list wrapper ('0')
    assert r.resolve('plain/bias') == 'plain/bias' + ck.VAR_SUFFIX       # not in the graph: name-based fallback
    assert r.resolve('encoder/blocks/1/kernel') is None


def test_object_paths_cover_the_registry():
    from strajnet_amd.modules import _param_spec
    paths = ck.object_paths(CFG)
    spec = _param_spec(CFG, 16, True, True)
    assert list(paths) == list(spec) and len(set(paths.values())) == len(paths)
    want = {
        'patch_embed_vecicle/proj/kernel': 'encoder/patch_embed_vecicle/proj/kernel',
        'flow_norm/gamma': 'encoder/flow_norm/gamma',
        'all_patch_norm/beta': 'encoder/all_patch_norm/beta',
        'flow_layers0/blocks1/attn/relative_position_bias_table': 'encoder/flow_layer/blocks/1/attn/relative_position_bias_table',
        'flow_layers0/downsample/reduction/kernel': 'encoder/flow_layer/downsample/reduction/kernel',
        'layers2/blocks0/mlp/fc2/bias': 'encoder/basic_layers/2/blocks/0/mlp/fc2/bias',
        'layers1/downsample/norm/gamma': 'encoder/basic_layers/1/downsample/norm/gamma',
        'fg_msa/proj_q/kernel': 'fg_msa_layer/proj_q/kernel',
        'fg_msa/warp_attn_rel_table': 'fg_msa_layer/rpe_table',
        'traj_net/traj_encoder/node_attention/query_kernel': 'trajnet_attn/traj_net/traj_encoder/node_attention/query_kernel',
        'traj_net/cross_attention/FFN1/kernel': 'trajnet_attn/traj_net/cross_attention/FFN1/kernel',
        'traj_net/seg_embed/kernel': 'trajnet_attn/traj_net/seg_embed/kernel',
        'cross_attn_obs5/mha/projection_bias': 'trajnet_attn/cross_attn_obs/5/mha/projection_bias',
        'cross_attn_obs0/norm2/gamma': 'trajnet_attn/cross_attn_obs/0/norm2/gamma',
        'decoder/upconv_3_0/kernel': 'decoder/upconv_0s/0/kernel',
        'decoder/upconv_0_0/bias': 'decoder/upconv_0s/3/bias',
        'decoder/resconv_2/kernel': 'decoder/res_layer/1/kernel',
        'decoder/resconv_f/bias': 'decoder/res_f/bias',
        'decoder/upconvf_1_0/kernel': 'decoder/upconv_f/0/kernel',
        'decoder/outconv/kernel': 'decoder/output_layer/kernel',
        'decoder/outconv_f/kernel': 'decoder/output_layer_f/kernel',
    }
    for k, v in want.items():
        assert paths[k] == v, (k, paths[k])
    deep = dict(CFG, depths=[2, 2, 6])
    assert ck.object_paths(deep)['layers2/blocks5/norm1/gamma'] == 'encoder/basic_layers/2/blocks/5/norm1/gamma'
    assert 'fg_msa/proj_q/kernel' not in ck.object_paths(CFG, fg_msa=False)


def test_model_checkpoint_roundtrip_on_the_registry(tmp_path):
    """All 13 277 788 scalars of cfg-256 through save_tf_checkpoint / load_tf_checkpoint, bit for bit."""
    from strajnet_amd.modules import _param_spec
    rng = np.random.default_rng(3)
    w = {n: rng.standard_normal(shape).astype(np.float32) for n, (shape, _) in _param_spec(CFG, 16, True, True).items()}
    assert sum(v.size for v in w.values()) == 13277788
    p = str(tmp_path / 'final_model.tf')
    ck.save_tf_checkpoint(p, w, CFG)
    assert os.path.getsize(p + '.data-00000-of-00001') > 4 * 13277788
    got = ck.load_tf_checkpoint(p, CFG)
    assert list(got) == list(w) and all(np.array_equal(got[n], w[n]) for n in w)
    r = ck.BundleReader(p)
    assert r.resolve('encoder/basic_layers/1/blocks/0/attn/qkv/kernel') == \
        'encoder/basic_layers/1/blocks/0/attn/qkv/kernel/.ATTRIBUTES/VARIABLE_VALUE'
    with pytest.raises(KeyError, match='not in the checkpoint'):          # a checkpoint of a smaller model must not half-load
        ck.load_tf_checkpoint(p, dict(CFG, depths=[2, 2, 6]))
    w.pop('decoder/outconv/bias')
    with pytest.raises(KeyError):
        ck.save_tf_checkpoint(str(tmp_path / 'x'), w, CFG)
