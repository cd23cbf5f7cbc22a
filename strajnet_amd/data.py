"""TFRecord input path of the reference (train.py:71-103, inference.py:67-96) without TensorFlow.

Host side (pure Python, no TF): TFRecord framing (length, masked CRC-32C, payload, masked CRC-32C) and the tf.Example
protobuf wire format restricted to what the reference's records contain -- one BytesList value per feature.
Device side: `decode_batch` uploads the raw feature bytes of a batch and expands them with one HIP launch per feature
(stj_decode_raw: bool / int8 / float32 / float64 -> float32, centre crop, scale), i.e. `_parse_image_function`.

    for batch in batches(read_tfrecord(path), 8):
        data = decode_batch(batch, 'cuda')          # dict: ogm [B,512,512,11,2], map_image [B,256,256,3], gt_flow [B,8,256,256,2] ...

A writer (`write_tfrecord`, `serialize_example`) exists for tests and synthetic data; it produces the same bytes TF reads.
"""
import ctypes
import struct

import numpy as np
import torch

from .ops import _p, _st, call

KIND = {'bool': 0, 'int8': 1, 'float32': 2, 'float64': 3}
ITEMSIZE = {'bool': 1, 'int8': 1, 'float32': 4, 'float64': 8}


def feature_spec(grid=512, out=256, test=False):
    """name -> (raw dtype, raw shape, (y0, x0, Ho, Wo) or None, scale).  train.py:87-103 (test=False) / inference.py:84-96."""
    c0 = (grid - out) // 2
    crop = (c0, c0, out, out)
    spec = {
        'centerlines': ('float64', (256, 10, 7), None, 1.0),
        'actors': ('float64', (48, 11, 8), None, 1.0),
        'occl_actors': ('float64', (16, 11, 8), None, 1.0),
        'ogm': ('bool', (grid, grid, 11, 2), None, 1.0),
        'map_image': ('int8', (out, out, 3), None, 1.0 / 256.0),
        'vec_flow': ('float32', (grid, grid, 2), None, 1.0),
    }
    if not test:
        spec.update({
            'gt_flow': ('float32', (8, grid, grid, 2), crop, 1.0),
            'origin_flow': ('float32', (8, grid, grid, 1), crop, 1.0),
            'gt_obs_ogm': ('bool', (8, grid, grid, 1), crop, 1.0),
            'gt_occ_ogm': ('bool', (8, grid, grid, 1), crop, 1.0),
        })
    return spec


# ---------------------------------------------------------------------------------------------------- CRC-32C (Castagnoli)
def _crc_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_TABLE = _crc_table()


def _crc32c_py(data, crc=0):
    c = crc ^ 0xFFFFFFFF
    for b in bytes(data):
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


_FAST = None


def crc32c(data, crc=0):
    """CRC-32C of `data` (bytes-like or ndarray), continuing from `crc`.  Uses the host routine of libstrajnet_hip.so
    (`stj_crc32c`, slice-by-8); the byte loop above is what it is tested against and what runs if the library is not built."""
    global _FAST
    if _FAST is None:
        try:
            from . import _lib
            _FAST = _lib.lib().stj_crc32c
        except Exception:
            _FAST = False
    if not _FAST:
        return _crc32c_py(data, crc)
    a = np.ascontiguousarray(data).reshape(-1).view(np.uint8) if isinstance(data, np.ndarray) else np.frombuffer(data, np.uint8)
    c = ctypes.c_uint(crc)
    if _FAST(a.ctypes.data if a.size else None, a.size, ctypes.byref(c)) != 0:
        raise RuntimeError('stj_crc32c failed')
    return c.value


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------- TFRecord framing
def read_tfrecord(path, check_data_crc=False):
    """Yield the payload of every record.  The 12-byte header CRC is always verified; the payload CRC (35 MB per example in
    pure Python) only on request."""
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise ValueError('truncated TFRecord header')
            n, hcrc = struct.unpack('<QI', head)
            if masked_crc(head[:8]) != hcrc:
                raise ValueError('corrupt TFRecord length')
            data = f.read(n)
            tail = f.read(4)
            if len(data) != n or len(tail) != 4:
                raise ValueError('truncated TFRecord payload')
            if check_data_crc and masked_crc(data) != struct.unpack('<I', tail)[0]:
                raise ValueError('corrupt TFRecord payload')
            yield data


def write_tfrecord(path, payloads):
    with open(path, 'wb') as f:
        for data in payloads:
            head = struct.pack('<Q', len(data))
            f.write(head + struct.pack('<I', masked_crc(head)) + data + struct.pack('<I', masked_crc(data)))


# ---------------------------------------------------------------------------------------------------- tf.Example (bytes features)
def _varint(buf, i):
    r, s = 0, 0
    while True:
        b = buf[i]
        i += 1
        r |= (b & 0x7F) << s
        if not b & 0x80:
            return r, i
        s += 7


def _fields(buf):
    """(field number, wire type, value) of a protobuf message; value = int (varint) or memoryview (length-delimited)."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = buf[i:i + ln]
            i += ln
        elif wt == 1:
            v = buf[i:i + 8]; i += 8
        elif wt == 5:
            v = buf[i:i + 4]; i += 4
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        yield fno, wt, v


def parse_example(payload):
    """tf.train.Example -> {feature name: bytes} for BytesList features (Example.features=1, Features.feature=1 (map entry:
    key=1, value=2), Feature.bytes_list=1, BytesList.value=1)."""
    buf = memoryview(payload)
    out = {}
    for fno, wt, feats in _fields(buf):
        if fno != 1 or wt != 2:
            continue
        for f2, w2, entry in _fields(feats):
            if f2 != 1 or w2 != 2:
                continue
            name, value = None, None
            for f3, w3, v in _fields(entry):
                if f3 == 1:
                    name = bytes(v).decode()
                elif f3 == 2:
                    for f4, w4, lst in _fields(v):
                        if f4 == 1 and w4 == 2:             # bytes_list
                            for f5, w5, b in _fields(lst):
                                if f5 == 1:
                                    value = b
            if name is not None and value is not None:
                out[name] = value
    return out


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(fno, payload):
    return _enc_varint((fno << 3) | 2) + _enc_varint(len(payload)) + payload


def serialize_example(features):
    """{name: bytes} -> tf.train.Example bytes (what data_preprocessing.py:365-381 writes with tf.train.Example)."""
    entries = b''
    for name in sorted(features):
        feat = _ld(1, _ld(1, bytes(features[name])))                  # Feature{bytes_list{value}}
        entries += _ld(1, _ld(1, name.encode()) + _ld(2, feat))       # map entry
    return _ld(1, entries)


def batches(examples, batch_size):
    buf = []
    for e in examples:
        buf.append(parse_example(e) if isinstance(e, (bytes, bytearray, memoryview)) else e)
        if len(buf) == batch_size:
            yield buf
            buf = []
    if buf:
        yield buf


# ---------------------------------------------------------------------------------------------------- device decode
def decode_batch(examples, device='cuda', grid=512, out=256, test=False):
    """examples: list of {feature: bytes} (parse_example output).  Returns float32 tensors on `device`, leading batch axis,
    shaped and cropped as _parse_image_function does (train.py:87-103): ogm [B,g,g,11,2], map_image [B,o,o,3] (/256),
    vec_flow [B,g,g,2], actors [B,48,11,8], occl_actors [B,16,11,8], centerlines [B,256,10,7] and, unless test,
    gt_flow [B,8,o,o,2], origin_flow / gt_obs_ogm / gt_occ_ogm [B,8,o,o,1]."""
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise RuntimeError('decode_batch: CUDA (ROCm) device only: the HIP path has no CPU fallback')
    B = len(examples)
    res = {}
    for name, (dtype, shape, crop, scale) in feature_spec(grid, out, test).items():
        nbytes = int(np.prod(shape)) * ITEMSIZE[dtype]
        host = torch.empty((B, nbytes), dtype=torch.uint8).pin_memory()
        for b, ex in enumerate(examples):
            raw = ex[name]
            if len(raw) != nbytes:
                raise ValueError(f'feature {name}: {len(raw)} bytes, expected {nbytes} for {dtype}{list(shape)}')
            host[b] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        src = host.to(dev, non_blocking=True)
        if crop is None:
            n_outer, H, W, C = B, 1, int(np.prod(shape)), 1
            y0, x0, Ho, Wo = 0, 0, 1, W
            oshape = (B,) + tuple(shape)
        else:
            n_outer, H, W, C = B * shape[0], shape[1], shape[2], shape[3]
            y0, x0, Ho, Wo = crop
            oshape = (B, shape[0], Ho, Wo, C)
        dst = torch.empty(oshape, dtype=torch.float32, device=dev)
        call('stj_decode_raw', _p(src), KIND[dtype], _p(dst), n_outer, H, W, C, y0, x0, Ho, Wo, float(scale), _st())
        res[name] = dst
    if test and 'scenario/id' in examples[0]:
        res['scenario/id'] = [bytes(ex['scenario/id']) for ex in examples]
    return res


# ---------------------------------------------------------------------------------------------------- per-step feed of a captured step
_COPY_STREAMS = {}


def _copy_stream(dev):
    """ONE upload stream per device for every HostFeed (streams are multiplexed on a few hardware queues in creation order; a fresh
    stream per feed lands on a different queue each time, and on the main chain's queue its SDMA barriers hold the step up)."""
    key = str(dev)
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(dev)
    return _COPY_STREAMS[key]


class HostFeed:
    """Feeds the static input tensors of a captured train step (graph.GraphedTrainStep.static) from PINNED host buffers, one batch
    per step (reference train.py:85-103,319: every step consumes a new batch), without stretching the step:

        feed = HostFeed(step.static, host, raw={'ogm': 'bool', ...})
        feed.start()                       # upload of the first batch
        for ...:
            feed.land()                    # device: wait for the upload, staging -> static inputs (copies / stj_decode_raw, ~0.1 ms)
                                           # host: the worker thread starts uploading what host[...] holds NOW (the next batch)
            step(); optimizer.step()
            feed.wait_uploaded()           # (a loader refills host[...] only after this returns)

    host[k]: pinned tensors -- float32, or for the keys of `raw` the TFRecord's own bytes ('bool' / 'int8', expanded on the device).
    Why a worker thread and 1.5 MB pieces (tools/probes/feed_probe.py, feed_probe3.py, feed_trace*.sh; B = 8: 77 MB of raw bytes per step):
      * hipMemcpyAsync of a large pinned buffer BLOCKS its host thread for the duration of the transfer.  Issued from the thread that
        replays the graph, no kernel runs until the last byte has arrived: the step grew by the full PCIe time (6.3 -> 8.4 ms),
        whichever stream / priority / order, also as a memcpy node inside the graph;
      * a kernel reading the pinned buffer over PCIe overlaps, but every kernel that runs beside it slows down 5-15x;
      * from a second thread the SDMA transfer overlaps the step -- except that while ONE call is blocked the replaying thread's
        launches stall too: a 33 MB tensor as one call left the GPU idle for 0.7 ms.  Measured by piece size (raw bytes, ms per step;
        resident inputs 6.28): 32 MB 7.42, 8 MB 7.65, 4 MB 7.49, 2 MB 6.60, 1.5 MB 6.47, 1 MB 6.57, 0.5 MB 6.48, 0.25 MB 6.48 --
        in 1.5 MB pieces the step keeps its resident-input time within 3 % (float32 host tensors, 141 MB: 7.2 ms)."""

    def __init__(self, static, host, raw=None, chunk_bytes=3 << 19):
        import threading
        self.static, self.host, self.raw = static, {k: h for k, h in host.items() if k in static}, dict(raw or {})
        dev = next(iter(static.values())).device
        self.dev = dev
        for k, h in self.host.items():
            if not h.is_pinned():
                raise ValueError(f'HostFeed: host[{k!r}] must be pinned memory')
        self.stage = {k: torch.empty(h.shape, dtype=h.dtype, device=dev) for k, h in self.host.items()}
        self.chunk = int(chunk_bytes)
        self.copy = _copy_stream(dev)
        self.up, self.landed = torch.cuda.Event(), torch.cuda.Event()
        self._go, self._enqueued = threading.Event(), threading.Event()
        self._stop = False
        self._err = None
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def _upload(self):
        with torch.cuda.stream(self.copy):
            self.copy.wait_event(self.landed)            # the staging buffers are free once the previous batch has left them
            for k, h in self.host.items():
                hs, ss = h.view(-1), self.stage[k].view(-1)
                step = max(1, self.chunk // h.element_size())
                for i in range(0, hs.numel(), step):
                    ss[i:i + step].copy_(hs[i:i + step], non_blocking=True)
            self.up.record(self.copy)

    def _run(self):
        torch.cuda.set_device(self.dev)
        while True:
            self._go.wait(); self._go.clear()
            if self._stop:
                return
            try:
                self._upload()
            except Exception as e:       # surfaced by the next land() / wait_uploaded()
                self._err = e
            self._enqueued.set()

    def _check(self):
        if self._err is not None:
            e, self._err = self._err, None
            raise e

    def start(self):
        self.landed.record(torch.cuda.current_stream(self.dev))
        self._go.set()

    def land(self):
        self._enqueued.wait(); self._enqueued.clear()       # the upload has been enqueued: its `up` event is recorded
        self._check()
        main = torch.cuda.current_stream(self.dev)
        main.wait_event(self.up)
        kind = {'bool': 0, 'int8': 1}
        for k, st in self.stage.items():
            dst = self.static[k]
            if k in self.raw:
                n = dst.numel()
                call('stj_decode_raw', _p(st), kind[self.raw[k]], _p(dst), 1, 1, n, 1, 0, 0, 1, n, (1.0 / 256.0) if self.raw[k] == 'int8' else 1.0, _st())
            else:
                dst.copy_(st, non_blocking=True)
        self.landed.record(main)
        self._go.set()                                       # next batch: whatever host[...] holds now

    def wait_uploaded(self):
        """Host-side: returns once the upload kicked off by the last land() has left the host buffers (a loader may refill them)."""
        self._enqueued.wait()
        self._check()
        self.up.synchronize()

    def close(self):
        self._stop = True
        self._go.set()
        self._thread.join(timeout=5)
