"""TF1 tensor-bundle checkpoint reader: on-disk structures checked against hand-assembled
bytes (independent of the writer) and by round trips through the minimal writer."""
import os
import struct

import numpy as np
import pytest

from spatialaudiogen_amd import checkpoint as ck
from spatialaudiogen_amd.weights import variable_specs, init_weights


def test_varints_and_protobuf_fields():
    for v in (0, 1, 127, 128, 300, 2 ** 32 + 5, 2 ** 63 - 1):
        enc = ck._put_varint(v)
        assert ck._varint(enc + b'\x99', 0) == (v, len(enc))
    # BundleEntryProto by hand: dtype=DT_FLOAT(1), shape {dim{size:3} dim{size:5}}, shard 0, offset 300, size 60, crc fixed32
    shape = b'\x12\x02\x08\x03' + b'\x12\x02\x08\x05'
    msg = b'\x08\x01' + b'\x12' + bytes([len(shape)]) + shape + b'\x20\xac\x02' + b'\x28\x3c' + b'\x35' + struct.pack('<I', 0xdeadbeef)
    e = ck._parse_entry(msg)
    assert (e['dtype'], e['shape'], e['offset'], e['size'], e['crc32c'], e['shard_id']) == (1, (3, 5), 300, 60, 0xdeadbeef, 0)


def test_prefix_compressed_block_by_hand():
    # entries: "audio/conv1/biases"->b"A", "audio/conv1/weights"->b"BB", "audio/conv2/w"->b"C"; one restart at 0
    e1 = bytes([0, 18, 1]) + b'audio/conv1/biases' + b'A'
    e2 = bytes([12, 7, 2]) + b'weights' + b'BB'             # shares "audio/conv1/"
    e3 = bytes([10, 3, 1]) + b'2/w' + b'C'                   # shares "audio/conv"
    block = e1 + e2 + e3 + struct.pack('<I', 0) + struct.pack('<I', 1)
    assert list(ck._block_entries(block)) == [(b'audio/conv1/biases', b'A'), (b'audio/conv1/weights', b'BB'), (b'audio/conv2/w', b'C')]


def test_snappy_decoder_literals_and_copies():
    # "abcdabcdabcdxyz": literal "abcd", copy(offset 4, len 8) with a 1-byte-offset tag, literal "xyz"
    stream = ck._put_varint(15) + bytes([(4 - 1) << 2]) + b'abcd' + bytes([((8 - 4) << 2) | 1, 4]) + bytes([(3 - 1) << 2]) + b'xyz'
    assert ck._snappy_decompress(stream) == b'abcdabcdabcdxyz'
    # 2-byte-offset copy
    stream = ck._put_varint(10) + bytes([(5 - 1) << 2]) + b'hello' + bytes([((5 - 1) << 2) | 2, 5, 0])
    assert ck._snappy_decompress(stream) == b'hellohello'


def test_crc32c_known_answers():
    assert ck.crc32c(b'') == 0
    assert ck.crc32c(b'123456789') == 0xE3069283
    assert ck.crc32c(b'\x00' * 32) == 0x8A9136AA


def test_round_trip_of_a_full_model_checkpoint(tmp_path):
    enc = ['audio', 'video']
    P = init_weights(variable_specs(enc), seed=11, mode='test')
    extra = {'step': np.array(150000, np.int64), 'beta1_power': np.array(0.9, np.float32)}
    for k in list(P)[:5]:
        extra[k + '/Adam'] = np.zeros_like(P[k])
        extra[k + '/Adam_1'] = np.ones_like(P[k])
    allv = dict(P); allv.update(extra)
    prefix = str(tmp_path / 'model.ckpt-150000')
    ck.save_checkpoint(prefix, allv, block_entries=16)           # several data blocks + prefix compression
    assert ck.latest_checkpoint(str(tmp_path)) == prefix
    entries, header = ck.read_index(prefix + '.index')
    assert header['num_shards'] == 1 and len(entries) == len(allv)
    assert entries['separation/deconv1/weights']['shape'] == (7, 16, 32, 64)
    got = ck.load_checkpoint(prefix)
    assert set(got) == set(allv)
    for k, v in allv.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    sub = ck.load_checkpoint(prefix, names={'localization/fc3/biases', 'step'})
    assert set(sub) == {'localization/fc3/biases', 'step'} and sub['step'] == 150000


def test_bad_files_are_rejected(tmp_path):
    p = tmp_path / 'x.index'
    p.write_bytes(b'\x00' * 100)
    with pytest.raises(ValueError):
        ck.read_index(str(p))
    assert ck.latest_checkpoint(str(tmp_path)) is None


def test_crc32c_known_answers_and_bulk_path():
    """crc32c (Castagnoli) known answers (RFC 3720 B.4) and the lane-parallel bulk routine against the scalar one."""
    assert ck.crc32c(b'123456789') == 0xE3069283
    assert ck.crc32c(bytes(32)) == 0x8A9136AA and ck.crc32c(b'\xff' * 32) == 0x62A8AB43
    r = np.random.default_rng(5)
    for n in (0, 1, 1000, 4 * 4096 - 1, 4 * 4096, 70001, 300000):
        data = r.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        assert ck.crc32c_bulk(data, lanes=256 if n < 200000 else 4096) == ck.crc32c(data), n


def test_writer_records_tensor_crcs_and_loader_checks_them(tmp_path):
    """BundleEntryProto.crc32c (field 6): written masked, as TF's BundleWriter does, and verified on load: a flipped payload
    byte must be reported, not loaded silently."""
    P = {'a/weights': np.arange(12, dtype=np.float32).reshape(3, 4), 'b/biases': np.linspace(-1, 1, 7).astype(np.float32)}
    prefix = str(tmp_path / 'model.ckpt-1')
    ck.save_checkpoint(prefix, P)
    entries, _ = ck.read_index(prefix + '.index')
    for name, arr in P.items():
        assert entries[name]['crc32c'] == ck._mask_crc(ck.crc32c(arr.tobytes())) != 0
    got = ck.load_checkpoint(prefix)
    assert all(np.array_equal(got[k], P[k]) for k in P)
    fn = prefix + '.data-00000-of-00001'
    raw = bytearray(open(fn, 'rb').read())
    raw[5] ^= 0x40
    open(fn, 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='crc32c'):
        ck.load_checkpoint(prefix)
    assert ck.load_checkpoint(prefix, verify=False)['a/weights'].shape == (3, 4)


def test_whole_bundle_assembled_by_hand(tmp_path):
    """Footer + metaindex + index block + several prefix-compressed data blocks + shard file written by tests/bundle_by_hand.py (an
    independent statement of the on-disk format with its own CRC32C and varint code; save_checkpoint is not involved), read through
    load_checkpoint with verification on; then one flipped payload byte must fail the per-tensor crc."""
    import bundle_by_hand as bh
    assert bh.crc32c(b'123456789') == 0xE3069283 and bh.crc32c(b'\x00' * 32) == 0x8A9136AA      # RFC 3720 known answers
    r = np.random.Generator(np.random.PCG64(3))
    V = {'audio_encoder/conv1/weights': r.normal(size=(7, 16, 1, 32)).astype(np.float32),
         'audio_encoder/conv1/biases': r.normal(size=(32,)).astype(np.float32),
         'audio_encoder/conv1/biases/Adam': np.zeros(32, np.float32),
         'video_encoder/conv2_1/conv_1/bn/gamma': r.uniform(0.5, 1.5, size=(64,)).astype(np.float32),
         'video_encoder/conv2_1/conv_1/bn/moving_variance': np.ones(64, np.float32),
         'localization/fc3/weights': r.normal(size=(512, 99)).astype(np.float32),
         'beta1_power': np.array(0.81, np.float32), 'step': np.array(150000, np.int64), 'x64': r.normal(size=(3, 5))}
    for i in range(40):                                            # enough keys for several data blocks / restart points
        V['separation/deconv%d/extra_%02d' % (i % 5 + 1, i)] = r.normal(size=(i % 7 + 1,)).astype(np.float32)
    prefix = str(tmp_path / 'model.ckpt-150000')
    bh.write_bundle(prefix, V, entries_per_block=7, restart_every=3, crc_below=1 << 30)
    assert ck.latest_checkpoint(str(tmp_path)) == prefix
    entries, header = ck.read_index(prefix + '.index')
    assert header['num_shards'] == 1 and set(entries) == set(V)
    assert entries['localization/fc3/weights']['shape'] == (512, 99) and entries['step']['shape'] == ()
    got = ck.load_checkpoint(prefix, verify=True)
    assert set(got) == set(V)
    for k in V:
        assert got[k].dtype == V[k].dtype and got[k].shape == V[k].shape and np.array_equal(got[k], V[k]), k
    only = ck.load_checkpoint(prefix, names={'audio_encoder/conv1/biases'})
    assert list(only) == ['audio_encoder/conv1/biases']
    fn = prefix + '.data-00000-of-00001'
    raw = bytearray(open(fn, 'rb').read())
    raw[entries['localization/fc3/weights']['offset'] + 17] ^= 0x40
    open(fn, 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='crc32c'):
        ck.load_checkpoint(prefix, verify=True)
