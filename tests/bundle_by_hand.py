"""A TF tensor-bundle (checkpoint V2) assembled from the published on-disk format, independently of the product's writer
(spatialaudiogen_amd/checkpoint.py: save_checkpoint is never used here): the fixture for "does load_checkpoint read a file it did
not write".  Format: <prefix>.index is a leveldb-style SSTable (prefix-compressed data blocks with restart arrays, 5-byte block
trailers = compression type + masked crc32c, an index block of BlockHandles, an empty metaindex block, 48-byte footer);
key "" holds BundleHeaderProto, every other key a BundleEntryProto; <prefix>.data-00000-of-00001 holds the raw little-endian
tensors back to back."""
import os
import struct

import numpy as np

DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}
MAGIC = 0xdb4775248b80fb57

_T = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (0x82F63B78 if _c & 1 else 0)
    _T.append(_c)


def crc32c(data):                       # Castagnoli, reflected, table-driven; RFC 3720 check value of b'123456789' is 0xE3069283
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = _T[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked(c):                          # leveldb / TF crc masking
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def field(num, wire, payload):
    return varint((num << 3) | wire) + payload


def entry_proto(a, offset, with_crc):
    dims = b''.join(field(2, 2, varint(len(d)) + d) for d in (field(1, 0, varint(int(n))) for n in a.shape))
    m = field(1, 0, varint(DT[a.dtype])) + field(2, 2, varint(len(dims)) + dims)
    if offset:
        m += field(4, 0, varint(offset))
    m += field(5, 0, varint(a.nbytes))
    if with_crc:
        m += field(6, 5, struct.pack('<I', masked(crc32c(a.tobytes()))))
    return m


def block(items, restart_every):
    out, restarts, prev = bytearray(), [], b''
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_every == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(prev)) and k[shared] == prev[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_bundle(prefix, variables, entries_per_block=24, restart_every=8, crc_below=1 << 16):
    """variables: {name: ndarray}.  Tensors of >= crc_below bytes carry no crc32c field (pure-Python CRC is too slow for 100 MB;
    the reader treats the field as optional)."""
    names = sorted(variables)
    header = field(1, 0, varint(1)) + field(2, 0, varint(0)) + field(3, 2, varint(2) + field(1, 0, varint(1)))
    items = [(b'', header)]
    off = 0
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for n in names:
            a = np.asarray(variables[n])                       # (np.ascontiguousarray would turn a scalar into shape (1,))
            a = a if a.flags['C_CONTIGUOUS'] else a.copy()
            items.append((n.encode(), entry_proto(a, off, a.nbytes < crc_below)))
            f.write(a.tobytes())
            off += a.nbytes
    body, index_items = bytearray(), []
    for i in range(0, len(items), entries_per_block):
        chunk = items[i:i + entries_per_block]
        blk = block(chunk, restart_every)
        handle = varint(len(body)) + varint(len(blk))
        body += blk + b'\x00' + struct.pack('<I', masked(crc32c(blk + b'\x00')))
        index_items.append((chunk[-1][0], handle))
    meta = block([], 1)
    mh = varint(len(body)) + varint(len(meta))
    body += meta + b'\x00' + struct.pack('<I', masked(crc32c(meta + b'\x00')))
    idx = block(index_items, 1)
    ih = varint(len(body)) + varint(len(idx))
    body += idx + b'\x00' + struct.pack('<I', masked(crc32c(idx + b'\x00')))
    foot = mh + ih
    body += foot + b'\x00' * (40 - len(foot)) + struct.pack('<Q', MAGIC)
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(body))
    with open(os.path.join(os.path.dirname(prefix), 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % ((os.path.basename(prefix),) * 2))
