"""TF1 checkpoint (tensor-bundle, "V2" format) reader — what `tf.train.Saver().restore` does for this path
(reference deploy.py:79-87, eval.py:98-118), without TensorFlow.

A checkpoint `<prefix>` is
  <prefix>.index                  an SSTable (LevelDB table format, tensorflow/core/lib/io/table) mapping
                                  "" -> BundleHeaderProto and tensor name -> BundleEntryProto
  <prefix>.data-00000-of-0000N    raw little-endian tensor bytes, addressed by (shard_id, offset, size)
and `<model_dir>/checkpoint` names the latest prefix (`model_checkpoint_path: "..."`).

`load_checkpoint(prefix)` returns {variable name: ndarray}; SptAudioGen.load_variables picks the names it
needs and ignores optimizer slots (`*/Adam`, `*/Adam_1`, `beta?_power`, `step`, `metrics/*`).

No real TF checkpoint exists offline, so the reader is validated against files produced by the minimal
writer below (same on-disk structures: prefix-compressed blocks, restart arrays, block handles, footer magic,
protobuf wire format) and against hand-assembled byte strings in tests/test_checkpoint.py.
"""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_LEN = 48
BLOCK_TRAILER = 5            # 1 byte compression type + 4 bytes masked crc32c

# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
DTYPE_IDS = {np.dtype(v): k for k, v in DTYPES.items()}


# ------------------------------------------------------------------------------------------------
# varints / protobuf wire format
# ------------------------------------------------------------------------------------------------
def _varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _proto_fields(buf):
    """Yield (field number, wire type, value) of one protobuf message (value: int or bytes)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]; pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + ln]); pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]; pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield field, wt, v


def _parse_shape(buf):
    """TensorShapeProto: repeated Dim dim = 2 { int64 size = 1; string name = 2; }; bool unknown_rank = 3."""
    dims = []
    for f, _, v in _proto_fields(buf):
        if f == 2:
            size = 0
            for f2, _, v2 in _proto_fields(v):
                if f2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
    return tuple(dims)


def _parse_entry(buf):
    """BundleEntryProto (tensorflow/core/protobuf/tensor_bundle.proto)."""
    e = {'dtype': 0, 'shape': (), 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': 0, 'slices': 0}
    for f, _, v in _proto_fields(buf):
        if f == 1: e['dtype'] = v
        elif f == 2: e['shape'] = _parse_shape(v)
        elif f == 3: e['shard_id'] = v
        elif f == 4: e['offset'] = v
        elif f == 5: e['size'] = v
        elif f == 6: e['crc32c'] = v
        elif f == 7: e['slices'] += 1
    return e


# ------------------------------------------------------------------------------------------------
# snappy (block compression type 1) — raw format decoder
# ------------------------------------------------------------------------------------------------
def _snappy_decompress(buf):
    n, pos = _varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]; pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little'); pos += nb
            ln += 1
            out += buf[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8); pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little'); pos += 4
        if off == 0 or off > len(out):
            raise ValueError('corrupt snappy stream')
        for _ in range(ln):                             # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy length mismatch')
    return bytes(out)


# ------------------------------------------------------------------------------------------------
# SSTable reading
# ------------------------------------------------------------------------------------------------
def _read_block(data, offset, size):
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype == 0:
        return raw
    if ctype == 1:
        return _snappy_decompress(raw)
    raise ValueError('unknown block compression type %d' % ctype)


def _block_entries(block):
    """Prefix-compressed entries: varint shared, varint non_shared, varint value_len, key delta, value; the
    block ends with uint32 restart offsets and their count."""
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        value = bytes(block[pos:pos + vlen]); pos += vlen
        yield key, value


def read_index(index_fn):
    """{tensor name: BundleEntryProto dict} and the header dict of a bundle .index file."""
    with open(index_fn, 'rb') as f:
        data = f.read()
    if len(data) < FOOTER_LEN or struct.unpack_from('<Q', data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError('%s is not a TF tensor-bundle index (bad table magic)' % index_fn)
    footer = data[-FOOTER_LEN:]
    _, pos = _varint(footer, 0)          # metaindex handle (offset, size) — unused
    _, pos = _varint(footer, pos)
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    entries, header = {}, {}
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size)):
        b_off, p = _varint(handle, 0)
        b_size, p = _varint(handle, p)
        for key, value in _block_entries(_read_block(data, b_off, b_size)):
            if key == b'':
                for f, _, v in _proto_fields(value):        # BundleHeaderProto: num_shards=1, endianness=2, version=3
                    if f == 1: header['num_shards'] = v
                    elif f == 2: header['endianness'] = v
            else:
                entries[key.decode('utf-8')] = _parse_entry(value)
    if header.get('endianness', 0) != 0:
        raise ValueError('big-endian bundles are not supported')
    return entries, header


def load_checkpoint(prefix, names=None, verify=True):
    """{name: ndarray} for every (or the requested) non-sliced tensor of checkpoint `prefix`; with `verify` the per-tensor
    crc32c of the index (when present) is checked."""
    entries, header = read_index(prefix + '.index')
    nshards = header.get('num_shards', 1)
    shards = {}
    out = {}
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e['slices']:
            raise ValueError('%s is a partitioned variable (slices): not supported' % name)
        if e['dtype'] not in DTYPES:
            continue                                     # strings etc.: nothing this path needs
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = np.memmap('%s.data-%05d-of-%05d' % (prefix, sid, nshards), dtype=np.uint8, mode='r')
        dt = np.dtype(DTYPES[e['dtype']])
        count = int(np.prod(e['shape'])) if e['shape'] else 1
        if count * dt.itemsize != e['size']:
            raise ValueError('%s: size %d does not match shape %s' % (name, e['size'], e['shape']))
        raw = shards[sid][e['offset']:e['offset'] + e['size']].tobytes()
        if verify and e['crc32c'] and _mask_crc(crc32c_bulk(raw)) != e['crc32c']:
            raise ValueError('%s: tensor bytes fail the crc32c recorded in the index (corrupt checkpoint)' % name)
        out[name] = np.frombuffer(raw, dtype=dt).reshape(e['shape']).copy()
    return out


def latest_checkpoint(model_dir):
    """tf.train.latest_checkpoint: parse `<model_dir>/checkpoint` (CheckpointState text proto)."""
    fn = os.path.join(model_dir, 'checkpoint')
    if not os.path.exists(fn):
        return None
    m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', open(fn).read())
    if not m:
        return None
    p = m.group(1)
    p = p if os.path.isabs(p) else os.path.join(model_dir, p)
    return p if os.path.exists(p + '.index') else None


# ------------------------------------------------------------------------------------------------
# minimal writer (tests / exporting synthetic weights in the reference's on-disk format)
# ------------------------------------------------------------------------------------------------
_CRC_TABLE = None


def crc32c(data):
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def _crc_tables():
    crc32c(b'')
    return np.asarray(_CRC_TABLE, dtype=np.uint32)


def crc32c_bulk(data, lanes=4096):
    """crc32c of a large buffer (tensor payloads: up to tens of MB) in O(len / lanes) numpy steps.  The CRC register is affine
    in its start value: R(chunk, r0) = R(chunk, 0) ^ Z_L(r0), Z_L = "process L zero bytes".  So the buffer is cut into `lanes`
    equal chunks whose registers (from 0) advance together, one byte position per numpy step, and are then folded left to right
    through Z_L (four 256-entry tables); head bytes that do not fill the lanes go through the scalar routine first."""
    buf = np.frombuffer(bytes(data) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).tobytes(), dtype=np.uint8)
    n = buf.size
    if n < 4 * lanes:
        return crc32c(buf.tobytes())
    T = _crc_tables()
    L = n // lanes
    head = n - L * lanes
    reg = 0xFFFFFFFF
    for b in buf[:head].tolist():                       # < lanes bytes
        reg = int(T[(reg ^ b) & 0xFF]) ^ (reg >> 8)
    body = buf[head:].reshape(lanes, L)
    r = np.zeros(lanes, dtype=np.uint32)
    for j in range(L):
        r = T[(r ^ body[:, j]) & np.uint32(0xFF)] ^ (r >> np.uint32(8))
    # Z_L as byte tables: advance the four single-byte basis sets through L zero bytes (vectorised over the 256 values)
    Z = []
    for byte in range(4):
        v = (np.arange(256, dtype=np.uint32) << np.uint32(8 * byte))
        for _ in range(L):
            v = T[v & np.uint32(0xFF)] ^ (v >> np.uint32(8))
        Z.append(v)
    for k in range(lanes):
        reg = int(Z[0][reg & 0xFF]) ^ int(Z[1][(reg >> 8) & 0xFF]) ^ int(Z[2][(reg >> 16) & 0xFF]) ^ int(Z[3][(reg >> 24) & 0xFF]) ^ int(r[k])
    return reg ^ 0xFFFFFFFF


def _pb(field, wt, payload):
    return _put_varint((field << 3) | wt) + payload


def _entry_proto(arr, offset):
    shape = b''.join(_pb(2, 2, _put_varint(len(d)) + d) for d in [_pb(1, 0, _put_varint(int(s))) for s in arr.shape])
    msg = _pb(1, 0, _put_varint(DTYPE_IDS[arr.dtype]))
    msg += _pb(2, 2, _put_varint(len(shape)) + shape)
    if offset:
        msg += _pb(4, 0, _put_varint(offset))
    msg += _pb(5, 0, _put_varint(arr.nbytes))
    # BundleEntryProto.crc32c (field 6, fixed32): masked crc32c of the tensor bytes - TF's BundleReader verifies it on restore
    msg += _pb(6, 5, struct.pack('<I', _mask_crc(crc32c_bulk(arr))))
    return msg


def _build_block(items, restart_interval=16):
    out, restarts, prev = bytearray(), [], b''
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts or [0]:
        out += struct.pack('<I', r)
    out += struct.pack('<I', max(len(restarts), 1))
    return bytes(out)


def save_checkpoint(prefix, variables, block_entries=64):
    """Write {name: ndarray} as <prefix>.index + <prefix>.data-00000-of-00001 and update `checkpoint`."""
    names = sorted(variables)
    offset = 0
    entries = [(b'', _pb(1, 0, _put_varint(1)) + _pb(3, 2, _put_varint(2) + _pb(1, 0, _put_varint(1))))]
    # atomic: both files are written under temporary names and renamed, the `checkpoint` state file is replaced last - a run
    # killed anywhere in here leaves `latest_checkpoint` pointing at the previous, complete bundle
    tmp_tag = '.tmp-%d' % os.getpid()
    with open(prefix + '.data-00000-of-00001' + tmp_tag, 'wb') as f:
        for n in names:
            a = np.asarray(variables[n])
            if a.ndim and not a.flags['C_CONTIGUOUS']:
                a = np.ascontiguousarray(a)
            entries.append((n.encode('utf-8'), _entry_proto(a, offset)))
            f.write(a.tobytes())
            offset += a.nbytes
        f.flush()
        os.fsync(f.fileno())
    out = bytearray()
    index_items = []
    for i in range(0, len(entries), block_entries):
        chunk = entries[i:i + block_entries]
        block = _build_block(chunk)
        handle = _put_varint(len(out)) + _put_varint(len(block))
        out += block + b'\x00' + struct.pack('<I', _mask_crc(crc32c(block + b'\x00')))
        index_items.append((chunk[-1][0] + b'\xff' if False else chunk[-1][0], handle))   # separator >= last key of the block
    meta = _build_block([])
    meta_handle = _put_varint(len(out)) + _put_varint(len(meta))
    out += meta + b'\x00' + struct.pack('<I', _mask_crc(crc32c(meta + b'\x00')))
    idx = _build_block(index_items, restart_interval=1)
    idx_handle = _put_varint(len(out)) + _put_varint(len(idx))
    out += idx + b'\x00' + struct.pack('<I', _mask_crc(crc32c(idx + b'\x00')))
    footer = meta_handle + idx_handle
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    out += footer
    with open(prefix + '.index' + tmp_tag, 'wb') as f:
        f.write(bytes(out))
        f.flush()
        os.fsync(f.fileno())
    os.replace(prefix + '.data-00000-of-00001' + tmp_tag, prefix + '.data-00000-of-00001')
    os.replace(prefix + '.index' + tmp_tag, prefix + '.index')
    state_fn = os.path.join(os.path.dirname(prefix) or '.', 'checkpoint')
    with open(state_fn + tmp_tag, 'w') as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % ((os.path.basename(prefix),) * 2))
    os.replace(state_fn + tmp_tag, state_fn)


def remove_checkpoint(prefix):
    """Delete the files of one bundle (tf.train.Saver(max_to_keep=1) drops the previous periodic checkpoint, train.py:176)."""
    for sfx in ('.index', '.data-00000-of-00001'):
        try:
            os.remove(prefix + sfx)
        except OSError:
            pass
