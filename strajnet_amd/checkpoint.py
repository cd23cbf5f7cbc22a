"""TensorFlow checkpoint ("tensor bundle") reader / writer and the Keras object-graph name map of the reference model.

Reference call sites: `model.save_weights('.../model_*.tf')` / `model.save_weights('.../final_model.tf')` (train.py:358,366),
`model.load_weights(path)` (train.py:372, inference.py:283, modules.py:874-876).  A `.tf` suffix selects Keras' TF-checkpoint
format: `<prefix>.index` (an SSTable of the LevelDB lineage: key -> BundleEntryProto) plus `<prefix>.data-00000-of-00001`
(the raw little-endian tensor bytes).  TensorFlow is not installed here, so the format is restated from its published
definition (tensorflow/core/util/tensor_bundle, lib/io/table, protobuf/tensor_bundle.proto, trackable_object_graph.proto) in
pure Python -- host logic, no device work.  FORMAT UNPINNED: no TF-written file exists in this image or in the reference
repository to check against; the tests pin reader and writer against each other and against hand-assembled byte vectors.

Variables are found the way TF's object-based restore finds them: by walking the `_CHECKPOINTABLE_OBJECT_GRAPH` from the root
along the Python attribute names of the reference classes (`encoder.basic_layers[1].blocks[0].attn.qkv.kernel`), not by the
Keras `name=` strings (two LayerNorms are both called `all_norm` and both output convs `outconv`, modules.py:517,557,701,726).
`object_paths()` is that attribute map for every tensor of `STrajNet`'s registry.
"""
import os
import struct
from collections import OrderedDict

import numpy as np

from .data import _varint, _fields, _enc_varint, _ld, crc32c

TABLE_MAGIC = 0xdb4775248b80fb57
OBJECT_GRAPH_KEY = '_CHECKPOINTABLE_OBJECT_GRAPH'
VAR_SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'
BLOCK_RESTART_INTERVAL = 16
BLOCK_SIZE = 256 << 10

# tensorflow/core/framework/types.proto DataType
DT = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 5: np.dtype('<i2'), 6: np.dtype('i1'),
      9: np.dtype('<i8'), 10: np.dtype('bool'), 17: np.dtype('<u2'), 19: np.dtype('<f2'), 22: np.dtype('<u4'),
      23: np.dtype('<u8')}
DT_STRING, DT_BFLOAT16 = 7, 14
DT_OF = {v: k for k, v in DT.items()}


def _mask(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _unmask(m):
    r = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------- snappy (block type 1)
def snappy_decompress(buf):
    """Raw snappy block decoder.  The bundle writer stores its index uncompressed; this covers tables written with the
    library default instead."""
    buf = bytes(buf)
    n, i = _varint(buf, 0)
    out = bytearray()
    while i < len(buf):
        tag = buf[i]; i += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[i:i + nb], 'little'); i += nb
            ln += 1
            out += buf[i:i + ln]; i += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[i]; i += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[i] | (buf[i + 1] << 8); i += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[i:i + 4], 'little'); i += 4
        if off == 0 or off > len(out):
            raise ValueError('snappy: bad copy offset')
        for _ in range(ln):                                   # overlapping copies repeat the pattern
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy: length mismatch')
    return bytes(out)


# ---------------------------------------------------------------------------------------------------- SSTable
def _read_block(buf, off, size, verify=True):
    """One table block: `size` content bytes, then 1 type byte and the masked CRC-32C of content + type."""
    raw = buf[off:off + size + 5]
    if len(raw) != size + 5:
        raise ValueError('table: truncated block')
    if verify and _mask(crc32c(raw[:size + 1])) != struct.unpack('<I', raw[size + 1:])[0]:
        raise ValueError('table: block checksum mismatch')
    kind = raw[size]
    body = bytes(raw[:size])
    if kind == 1:
        body = snappy_decompress(body)
    elif kind != 0:
        raise ValueError(f'table: unknown block compression {kind}')
    return body


def _block_entries(body):
    """(key, value) of a block, undoing the shared-prefix key compression."""
    nrestart = struct.unpack('<I', body[-4:])[0]
    end = len(body) - 4 - 4 * nrestart
    if end < 0:
        raise ValueError('table: bad restart array')
    i, key = 0, b''
    while i < end:
        shared, i = _varint(body, i)
        fresh, i = _varint(body, i)
        vlen, i = _varint(body, i)
        if shared > len(key) or i + fresh + vlen > end:
            raise ValueError('table: corrupt entry')
        key = key[:shared] + body[i:i + fresh]
        i += fresh
        yield key, body[i:i + vlen]
        i += vlen


def _handle(buf, i):
    off, i = _varint(buf, i)
    size, i = _varint(buf, i)
    return off, size, i


def read_table(buf, verify=True):
    """All (key bytes, value bytes) of an SSTable, in key order."""
    if len(buf) < 48 or struct.unpack('<Q', buf[-8:])[0] != TABLE_MAGIC:
        raise ValueError('not a TensorFlow checkpoint index (bad table magic)')
    foot = bytes(buf[-48:])
    _, _, i = _handle(foot, 0)                                  # metaindex (unused by the bundle)
    ioff, isize, _ = _handle(foot, i)
    out = []
    for _, hv in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, bsize, _ = _handle(hv, 0)
        out.extend(_block_entries(_read_block(buf, boff, bsize, verify)))
    return out


def _build_block(items, interval=BLOCK_RESTART_INTERVAL):
    body, restarts, prev = bytearray(), [], b''
    for n, (k, v) in enumerate(items):
        shared = 0
        if n % interval == 0:
            restarts.append(len(body))
        else:
            m = min(len(prev), len(k))
            while shared < m and prev[shared] == k[shared]:
                shared += 1
        body += _enc_varint(shared) + _enc_varint(len(k) - shared) + _enc_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack('<I', r)
    body += struct.pack('<I', len(restarts))
    return bytes(body)


def write_table(items, block_size=BLOCK_SIZE):
    """Serialise sorted (key, value) pairs: data blocks, an empty metaindex block, the index block and the 48-byte footer.
    Blocks are stored uncompressed (type 0), as the bundle writer does."""
    items = list(items)
    if any(items[i][0] >= items[i + 1][0] for i in range(len(items) - 1)):
        raise ValueError('table keys must be strictly increasing')
    out = bytearray()

    def emit(body):
        off = len(out)
        out.extend(body + b'\0')
        out.extend(struct.pack('<I', _mask(crc32c(body + b'\0'))))
        return _enc_varint(off) + _enc_varint(len(body))

    index, cur, cur_bytes = [], [], 0
    for kv in items:
        cur.append(kv)
        cur_bytes += len(kv[0]) + len(kv[1]) + 3
        if cur_bytes >= block_size:
            index.append((cur[-1][0], emit(_build_block(cur))))
            cur, cur_bytes = [], 0
    if cur or not index:
        index.append((cur[-1][0] if cur else b'', emit(_build_block(cur))))
    meta = emit(_build_block([]))
    idx = emit(_build_block(index, interval=1))
    foot = meta + idx
    out.extend(foot + b'\0' * (40 - len(foot)) + struct.pack('<Q', TABLE_MAGIC))
    return bytes(out)


# ---------------------------------------------------------------------------------------------------- bundle protos
def _parse_entry(buf):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'slices': 0}
    for fno, wt, v in _fields(memoryview(bytes(buf))):
        if fno == 1:
            e['dtype'] = v
        elif fno == 2:
            for f2, _, v2 in _fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            size = v3
                    e['shape'].append(size)
        elif fno == 3:
            e['shard_id'] = v
        elif fno == 4:
            e['offset'] = v
        elif fno == 5:
            e['size'] = v
        elif fno == 6:
            e['crc32c'] = struct.unpack('<I', bytes(v))[0]
        elif fno == 7:
            e['slices'] += 1
    return e


def _vint(fno, v):
    return _enc_varint(fno << 3) + _enc_varint(v)


def _entry_bytes(dtype, shape, offset, size, crc):
    dims = b''.join(_ld(2, _vint(1, int(d))) for d in shape)
    out = _vint(1, dtype) + _ld(2, dims)
    if offset:
        out += _vint(4, offset)
    out += _vint(5, size) + _enc_varint((6 << 3) | 5) + struct.pack('<I', crc)
    return out


def _header_bytes(num_shards=1):
    return _vint(1, num_shards) + _ld(3, _vint(1, 1))          # endianness LITTLE (0, default), version.producer = 1


def parse_object_graph(buf):
    """TrackableObjectGraph -> list of nodes {'children': {local_name: node_id}, 'attributes': [(name, full_name, key)]}."""
    nodes = []
    for fno, _, v in _fields(memoryview(bytes(buf))):
        if fno != 1:
            continue
        node = {'children': OrderedDict(), 'attributes': []}
        for f2, _, v2 in _fields(v):
            if f2 == 1:
                nid, name = 0, ''
                for f3, _, v3 in _fields(v2):
                    if f3 == 1:
                        nid = v3
                    elif f3 == 2:
                        name = bytes(v3).decode()
                node['children'][name] = nid
            elif f2 == 2:
                a = ['', '', '']
                for f3, _, v3 in _fields(v2):
                    if 1 <= f3 <= 3:
                        a[f3 - 1] = bytes(v3).decode()
                node['attributes'].append(tuple(a))
        nodes.append(node)
    return nodes


def build_object_graph(var_paths):
    """TrackableObjectGraph bytes of a tree whose leaves are the variables at `var_paths` ('a/b/0/kernel'); each leaf carries
    the VARIABLE_VALUE attribute with checkpoint key '<path>/.ATTRIBUTES/VARIABLE_VALUE', as Keras writes it."""
    nodes = [{'children': OrderedDict(), 'attr': None}]
    for path in var_paths:
        cur = 0
        for part in path.split('/'):
            nxt = nodes[cur]['children'].get(part)
            if nxt is None:
                nxt = len(nodes)
                nodes[cur]['children'][part] = nxt
                nodes.append({'children': OrderedDict(), 'attr': None})
            cur = nxt
        nodes[cur]['attr'] = path
    out = b''
    for n in nodes:
        body = b''
        for name, nid in n['children'].items():
            body += _ld(1, (_vint(1, nid) if nid else b'') + _ld(2, name.encode()))
        if n['attr'] is not None:
            body += _ld(2, _ld(1, b'VARIABLE_VALUE') + _ld(2, n['attr'].encode()) + _ld(3, (n['attr'] + VAR_SUFFIX).encode()))
        out += _ld(1, body)
    return out


# ---------------------------------------------------------------------------------------------------- reader / writer
class BundleReader:
    """`BundleReader(prefix)`: keys(), has(key), entry(key), get(key) -> ndarray (bytes for a scalar string), object_graph(),
    resolve(path) -> checkpoint key of the variable reached from the root along an attribute path."""

    def __init__(self, prefix, verify=True):
        self.prefix = prefix
        self.verify = verify
        with open(prefix + '.index', 'rb') as f:
            items = read_table(f.read(), verify)
        self.entries = OrderedDict()
        self.num_shards = 1
        for k, v in items:
            if k == b'':
                for fno, _, val in _fields(memoryview(bytes(v))):
                    if fno == 1:
                        self.num_shards = val
                    elif fno == 2 and val != 0:
                        raise ValueError('big-endian checkpoint bundles are not supported')
                continue
            self.entries[k.decode()] = _parse_entry(v)
        self._graph = None
        self._shards = {}

    def keys(self):
        return list(self.entries)

    def has(self, key):
        return key in self.entries

    def entry(self, key):
        return self.entries[key]

    def _shard(self, sid):
        if sid not in self._shards:
            self._shards[sid] = np.memmap(f'{self.prefix}.data-{sid:05d}-of-{self.num_shards:05d}', dtype=np.uint8, mode='r')
        return self._shards[sid]

    def get(self, key):
        e = self.entries[key]
        if e['slices']:
            raise ValueError(f'{key}: partitioned (sliced) variables are not supported')
        raw = self._shard(e['shard_id'])[e['offset']:e['offset'] + e['size']]
        if len(raw) != e['size']:
            raise ValueError(f'{key}: data shard is truncated')
        if e['dtype'] == DT_STRING:
            return self._strings(key, e, bytes(raw))
        if e['dtype'] == DT_BFLOAT16:
            a = (np.frombuffer(raw, '<u2').astype(np.uint32) << 16).view(np.float32)
        elif e['dtype'] in DT:
            a = np.frombuffer(raw, DT[e['dtype']])
        else:
            raise ValueError(f'{key}: unsupported dtype enum {e["dtype"]}')
        if self.verify and e['crc32c'] is not None and _mask(crc32c(raw)) != e['crc32c']:
            raise ValueError(f'{key}: tensor checksum mismatch')
        if a.size != int(np.prod(e['shape'], dtype=np.int64)):
            raise ValueError(f'{key}: {a.size} elements for shape {e["shape"]}')
        return a.reshape(e['shape'])

    def _strings(self, key, e, raw):
        """[varint64 length]*n, the masked CRC-32C of the lengths (4 bytes), then the bytes back to back.  Checksums follow
        tensor_bundle ReadStringTensor: lengths enter the running CRC as uint32 (uint64 above 2^32-1)."""
        n = int(np.prod(e['shape'], dtype=np.int64))
        i, lens = 0, []
        c = 0
        for _ in range(n):
            ln, i = _varint(raw, i)
            lens.append(ln)
            c = crc32c(_len_word(ln), c)
        lcs = raw[i:i + 4]
        if self.verify and (len(lcs) != 4 or struct.unpack('<I', lcs)[0] != _mask(c)):
            raise ValueError(f'{key}: string length checksum mismatch')
        c = crc32c(lcs, c)
        i += 4
        out = []
        for ln in lens:
            out.append(raw[i:i + ln]); i += ln
            c = crc32c(out[-1], c)
        if i != len(raw):
            raise ValueError(f'{key}: string tensor size mismatch')
        if self.verify and e['crc32c'] is not None and _mask(c) != e['crc32c']:
            raise ValueError(f'{key}: string tensor checksum mismatch')
        return out[0] if not e['shape'] else np.array(out, dtype=object).reshape(e['shape'])

    def object_graph(self):
        if self._graph is None:
            self._graph = parse_object_graph(self.get(OBJECT_GRAPH_KEY)) if self.has(OBJECT_GRAPH_KEY) else []
        return self._graph

    def resolve(self, path):
        """Checkpoint key of the variable at attribute path 'a/b/0/kernel', or None."""
        g = self.object_graph()
        if g:
            cur = 0
            for part in path.split('/'):
                cur = g[cur]['children'].get(part)
                if cur is None or cur >= len(g):
                    break
            else:
                for name, _, key in g[cur]['attributes']:
                    if name == 'VARIABLE_VALUE' and key in self.entries:
                        return key
        key = path + VAR_SUFFIX                                 # name-based checkpoints / graphs that miss the path
        return key if key in self.entries else None


def _len_word(n):
    """A string length as it enters the bundle checksum: fixed uint32 when it fits, else uint64 (tensor_bundle.cc)."""
    return struct.pack('<I', n) if n <= 0xFFFFFFFF else struct.pack('<Q', n)


class BundleWriter:
    """`w = BundleWriter(prefix); w.add(key, array); w.add_string(key, bytes); w.finish()` -> `<prefix>.index` +
    `<prefix>.data-00000-of-00001`."""

    def __init__(self, prefix):
        self.prefix = prefix
        d = os.path.dirname(prefix)
        if d:
            os.makedirs(d, exist_ok=True)
        self._data = open(prefix + '.data-00000-of-00001', 'wb')
        self._off = 0
        self._entries = {}

    def _put(self, key, dtype, shape, raw, crc):
        if key in self._entries or key == '':
            raise ValueError(f'duplicate / empty checkpoint key {key!r}')
        self._data.write(raw)
        self._entries[key] = _entry_bytes(dtype, shape, self._off, len(raw), crc)
        self._off += len(raw)

    def add(self, key, array):
        a = np.asarray(array)
        dt = a.dtype.newbyteorder('<') if a.dtype.byteorder == '>' else a.dtype
        if np.dtype(dt) not in DT_OF:
            raise ValueError(f'{key}: dtype {a.dtype} has no TensorFlow DataType here')
        raw = a.astype(dt, copy=False).tobytes()                # C order
        self._put(key, DT_OF[np.dtype(dt)], a.shape, raw, _mask(crc32c(raw)))

    def add_string(self, key, value):
        """Scalar DT_STRING (tensor_bundle WriteStringTensor): varint64 length, masked CRC-32C of the length, then the bytes.  The
        running checksum is extended with the length as a uint32 when it fits 32 bits (a uint64 only above that), then with the
        4 bytes of the masked length checksum, then with the string bytes; the entry crc32c is its masked value."""
        value = bytes(value)
        c = crc32c(_len_word(len(value)))
        lcs = struct.pack('<I', _mask(c))
        c = crc32c(value, crc32c(lcs, c))
        self._put(key, DT_STRING, (), _enc_varint(len(value)) + lcs + value, _mask(c))

    def finish(self):
        self._data.close()
        items = [(b'', _header_bytes())] + [(k.encode(), v) for k, v in sorted(self._entries.items(), key=lambda kv: kv[0].encode())]
        with open(self.prefix + '.index', 'wb') as f:
            f.write(write_table(items))


# ---------------------------------------------------------------------------------------------------- reference name map
def object_paths(cfg, fg_msa=True, fg=True):
    """registry name (strajnet_amd.modules._param_spec) -> attribute path from the reference's `STrajNet` root.

    STrajNet: encoder / trajnet_attn / fg_msa_layer / decoder (modules.py:782-800).  Encoder: patch_embed_*, flow_norm,
    flow_layer, basic_layers[L], all_patch_norm (modules.py:490-557); BasicLayer.blocks[i] / .downsample (:329,341); block:
    norm1, attn.{qkv,relative_position_bias_table,proj}, norm2, mlp.{fc1,fc2} (:179-187,76-83,36-37).  TrajNetCrossAttention:
    traj_net.{traj_encoder,cross_attention,obs_norm,occ_norm,seg_embed}, cross_attn_obs[t] (trajNet.py:101-120,241,257).
    Decoder (shallow_decode = 1, decode_inds [3,2,1,0]): upconv_0s[j] = upconv_{3-j}_0, res_layer[j] = resconv_{3-j},
    upconv_f[j] = upconvf_{1-j}_0, res_f, output_layer, output_layer_f (modules.py:634,667-730)."""
    from .modules import _param_spec
    names = list(_param_spec(cfg, 16, fg_msa, fg))          # the FG-MSA bottleneck is 16x16 in every configuration (modules.py:799)
    fixed = {'decoder/upconv_3_0': 'decoder/upconv_0s/0', 'decoder/upconv_2_0': 'decoder/upconv_0s/1',
             'decoder/upconv_1_0': 'decoder/upconv_0s/2', 'decoder/upconv_0_0': 'decoder/upconv_0s/3',
             'decoder/resconv_3': 'decoder/res_layer/0', 'decoder/resconv_2': 'decoder/res_layer/1',
             'decoder/resconv_f': 'decoder/res_f', 'decoder/upconvf_1_0': 'decoder/upconv_f/0',
             'decoder/upconvf_0_0': 'decoder/upconv_f/1', 'decoder/outconv': 'decoder/output_layer',
             'decoder/outconv_f': 'decoder/output_layer_f', 'fg_msa/warp_attn_rel_table': 'fg_msa_layer/rpe_table'}
    out = OrderedDict()
    for n in names:
        head, leaf = n.rsplit('/', 1)
        if n in fixed:
            p = fixed[n]
        elif head in fixed:
            p = f'{fixed[head]}/{leaf}'
        elif n.startswith('fg_msa/'):
            p = 'fg_msa_layer/' + n[len('fg_msa/'):]
        elif n.startswith('traj_net/'):
            p = 'trajnet_attn/' + n
        elif n.startswith('cross_attn_obs'):
            i, rest = n[len('cross_attn_obs'):].split('/', 1)
            p = f'trajnet_attn/cross_attn_obs/{i}/{rest}'
        else:                                                   # encoder
            parts = n.split('/')
            if parts[0] == 'flow_layers0':
                parts[0:1] = ['flow_layer']
            elif parts[0].startswith('layers'):
                parts[0:1] = ['basic_layers', parts[0][len('layers'):]]
            parts = [q for part in parts for q in ((['blocks', part[6:]]) if part.startswith('blocks') and part[6:].isdigit() else [part])]
            p = 'encoder/' + '/'.join(parts)
        out[n] = p
    if len(set(out.values())) != len(out):
        raise AssertionError('object path map is not injective')
    return out


def load_tf_checkpoint(prefix, cfg, fg_msa=True, fg=True, verify=True):
    """Read every registry tensor out of a reference checkpoint -> {registry name: f32 ndarray}.  Raises KeyError naming the
    variables that cannot be found (nothing is silently left at its initial value)."""
    r = BundleReader(prefix, verify)
    out, missing = OrderedDict(), []
    for name, path in object_paths(cfg, fg_msa, fg).items():
        key = r.resolve(path)
        if key is None:
            missing.append(path)
            continue
        out[name] = np.asarray(r.get(key), dtype=np.float32)
    if missing:
        raise KeyError(f'{prefix}: {len(missing)} variables not in the checkpoint, e.g. {missing[:4]}')
    return out


def save_tf_checkpoint(prefix, weights, cfg, fg_msa=True, fg=True):
    """Write {registry name: array} as a TF-format checkpoint whose object graph carries the reference's attribute paths."""
    paths = object_paths(cfg, fg_msa, fg)
    missing = [n for n in paths if n not in weights]
    if missing:
        raise KeyError(f'save_tf_checkpoint: missing {missing[:4]}')
    w = BundleWriter(prefix)
    for name, path in paths.items():
        w.add(path + VAR_SUFFIX, np.asarray(weights[name], dtype=np.float32))
    w.add_string(OBJECT_GRAPH_KEY, build_object_graph(paths.values()))
    w.finish()
