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


def test_library_matches_the_sources_it_travels_with():
    """The .so is git-ignored and travels prebuilt: its link-time source digest must equal the digest of csrc/ + include/sagen.h in
    this tree (a stale binary would make every parity test a statement about other sources)."""
    from spatialaudiogen_amd import _lib, build
    assert _lib.lib().sagen_source_digest().decode() == build.source_digest()


def _kernel_bodies(text):
    """{mangled symbol: disassembly} of every function in the objdump listing."""
    out, name = {}, None
    for line in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <([^>]+)>:', line)
        if m:
            name = m.group(1)
            out[name] = []
        elif name is not None:
            out[name].append(line)
    return {k: '\n'.join(v) for k, v in out.items()}


def test_wave_reductions_use_dpp_not_the_lds_crossbar(disassembly):
    """north_star: "wavefront-level DPP reductions for the per-bin magnitude and the ambisonic power map".  The reductions of the
    power map, the evaluation metrics, the mask / iSTFT energy sums and the stft loss go through wave_reduce.h (DPP row controls
    + v_permlane{16,32}_swap); none of those kernels may fall back to ds_bpermute_b32 (what __shfl_xor compiles to)."""
    text, _ = disassembly
    bodies = _kernel_bodies(text)
    wanted = ('power_moments_kernel', 'power_moments_batched_kernel', 'eval_time_kernel', 'eval_lsd_reduce_kernel',
              'stft_loss_grad_kernel', 'mask_istft_kernel', 'mask_istft_bwd_kernel', 'mask_bwd_finalize_kernel')
    seen, with_dpp = set(), set()
    for sym, body in bodies.items():
        for w in wanted:
            if w in sym:
                seen.add(w)
                assert 'ds_bpermute' not in body and 'ds_swizzle' not in body, '%s reduces through the LDS crossbar' % sym
                if re.search(r'_dpp|permlane(16|32)_swap', body):    # (an instantiation may need no cross-lane step at all: mask_istft_kernel<16>)
                    with_dpp.add(w)
    missing = set(wanted) - seen
    assert not missing, 'kernels not found in the code object: %s' % sorted(missing)
    assert with_dpp == set(wanted), 'no DPP / permlane instruction in any instantiation of: %s' % sorted(set(wanted) - with_dpp)
    assert 'ds_bpermute' not in text, 'some kernel still shuffles through ds_bpermute_b32'
