"""Guards on the BUILT code object (no GPU needed): the gfx950 kernels inside libsagen_hip.so contain no packed-fp32 VALU
instruction and use no scratch.

Why: on MI355X a wave executing v_pk_add/mul/fma_f32 can get wrong results while a wave of another kernel issues
v_mfma_f32_32x32x16_bf16 on the same SIMD - seen as corrupted STFTs when two forwards are in flight.  The library is built
with -fno-slp-vectorize -fno-vectorize for that reason; a compiler or flag change that silently brings the packed ops back
must fail a test, not corrupt audio."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def device_code_objects(lib_path, tmp):
    """The gfx950 ELF images of every translation unit: .hip_fatbin holds one clang offload bundle per TU."""
    blob = os.path.join(tmp, 'fatbin')
    subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + blob, lib_path, os.path.join(tmp, 'copy.so')])
    data = open(blob, 'rb').read()
    out, pos = [], data.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from('<Q', data, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from('<QQQ', data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if 'gfx950' in triple and size:
                fn = os.path.join(tmp, 'dev%d.co' % len(out))
                open(fn, 'wb').write(data[pos + off:pos + off + size])
                out.append(fn)
        pos = data.find(MAGIC, pos + len(MAGIC))
    return out


@pytest.fixture(scope='module')
def disassembly(tmp_path_factory):
    from spatialaudiogen_amd import build
    lib = build.build(verbose=False)
    tmp = str(tmp_path_factory.mktemp('co'))
    objs = device_code_objects(lib, tmp)
    assert len(objs) >= 8, 'expected one gfx950 code object per HIP source with kernels (%d found)' % len(objs)
    text = ''.join(subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', o], capture_output=True, text=True, check=True).stdout for o in objs)
    notes = ''.join(subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', o], capture_output=True, text=True, check=True).stdout for o in objs)
    return text, notes


def test_no_packed_fp32_valu_in_any_kernel(disassembly):
    text, _ = disassembly
    assert text.count('v_mfma_f32_32x32x16_bf16') > 1000           # the right code objects were found
    bad = re.findall(r'\bv_pk_(?:add|mul|fma)_f32\b', text)
    assert not bad, '%d packed-fp32 VALU instructions in the built kernels (first: %s): rebuild with -fno-slp-vectorize -fno-vectorize' % (len(bad), bad[0])


def test_no_scratch_and_no_spills(disassembly):
    _, notes = disassembly
    priv = [int(v) for v in re.findall(r'\.private_segment_fixed_size:\s*(\d+)', notes)]
    spills = [int(v) for v in re.findall(r'\.vgpr_spill_count:\s*(\d+)', notes)]
    assert len(priv) >= 50
    assert max(priv) == 0 and max(spills + [0]) == 0, 'a kernel uses scratch memory / spills registers'


def test_library_reports_its_build_flags():
    """The flags are compiled into the library (sagen_build_info) and _lib.lib() refuses a library built without the two
    vectoriser switches; the staleness check of build.py hashes the flags, so a flag change rebuilds every object."""
    from spatialaudiogen_amd import _lib, build
    info = _lib.lib().sagen_build_info().decode()
    for flag in ('-fno-slp-vectorize', '-fno-vectorize', '--offload-arch=gfx950'):
        assert flag in info, info
    assert build.flags_digest() in open(os.path.join(build.OBJ, 'flags.stamp')).read()
