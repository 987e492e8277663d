"""Build libsagen_hip.so (hand-written HIP, gfx950 only) in-tree with hipcc.

    python -m spatialaudiogen_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to csrc/build/ (git-ignored); the shared library
lands next to this file so it travels with the source tree to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libsagen_hip.so')
SOURCES = ['conv3p.hip', 'conv3h.hip', 'conv3g.hip', 'p3.hip', 'igemm3dw.hip', 'igemm3s2.hip', 'stempool.hip', 'stem8.hip', 'igemm.hip', 'fcm.hip', 'igemm3.hip', 'elementwise.hip', 'fft.hip', 'eval.hip', 'train.hip', 'wgrad.hip', 'wgrad3h.hip', 'backward.hip', 'model.hip', 'train_model.hip', 'api.hip']
# every header of csrc/ (the listing source_digest() hashes) + the public one: editing any of them rebuilds every object
HEADERS = sorted(os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith('.h')) + \
          [os.path.join(os.path.dirname(os.path.dirname(CSRC)), 'include', 'sagen.h')]
# -fno-slp-vectorize -fno-vectorize: no packed-fp32 VALU (v_pk_add/mul/fma_f32).  Measured on MI355X: a wave executing packed-fp32 ops gives
# wrong results while a wave of another kernel issues v_mfma_f32_32x32x16_bf16 on the same SIMD (the LDS FFT kernels next to the
# bf16x3 contractions of another stream; tools/victims/) - and packed fp32 is no faster next to MFMAs anyway.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-fno-slp-vectorize', '-fno-vectorize']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def flags_digest():
    """Hash of everything besides the sources that decides what the objects contain: the flags and this file."""
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    h.update(open(os.path.abspath(__file__), 'rb').read())
    return h.hexdigest()[:16]


def source_digest():
    """sha256[:16] over the library's sources (name + contents, sorted): compiled into the library at link time
    (sagen_source_digest) so that a stale prebuilt .so can be told from the tree it travels with."""
    import hashlib
    files = sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.h')))
    h = hashlib.sha256()
    for f in files:
        h.update(f.encode() + b'\0' + open(os.path.join(CSRC, f), 'rb').read() + b'\0')
    h.update(b'sagen.h\0' + open(HEADERS[-1], 'rb').read())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    stamp = os.path.join(OBJ, 'flags.stamp')
    if not os.path.exists(stamp) or open(stamp).read().strip() != flags_digest():
        force = True                    # objects built with other flags (or by another build.py) are never reused
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + HEADERS):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ['-DSAGEN_BUILD_FLAGS="%s"' % ' '.join(FLAGS), '-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (s, r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, max(1, len(jobs)))) as ex:
        for err in ex.map(compile_one, jobs):
            if verbose and err.strip():
                print(err)
    objs = [os.path.join(OBJ, s.replace('.hip', '.o')) for s in SOURCES]
    # the digest object: one host-only translation unit, regenerated whenever the digest of the sources changes
    digest = source_digest()
    dsrc, dobj = os.path.join(OBJ, 'source_digest.cpp'), os.path.join(OBJ, 'source_digest.o')
    text = 'extern "C" const char* sagen_source_digest(void) { return "%s"; }\n' % digest
    if force or not os.path.exists(dsrc) or open(dsrc).read() != text or not os.path.exists(dobj):
        with open(dsrc, 'w') as f:
            f.write(text)
        r = subprocess.run([hipcc, '-O1', '-fPIC', '-c', '-x', 'c++', dsrc, '-o', dobj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('digest object failed:\n%s' % r.stderr)
    objs.append(dobj)
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stderr)
    with open(stamp, 'w') as f:
        f.write(flags_digest() + '\n')
    return LIB


CPU_TWIN_SRC = os.path.join(HERE, 'csrc_cpu', 'sagen_cpu.cpp')
CPU_TWIN_LIB = os.path.join(HERE, 'libsagen_cpu.so')


def build_cpu_twin(force=False):
    """libsagen_cpu.so: the op level of include/sagen.h in plain C++ on host pointers (csrc_cpu/sagen_cpu.cpp) - test infrastructure
    for a container without a GPU (SAGEN_LIB=<this file> python -m pytest tests/test_gpu_ops.py ...), never a fallback."""
    if force or _stale(CPU_TWIN_LIB, [CPU_TWIN_SRC, HEADERS[-1]]):
        cxx = os.environ.get('CXX', 'g++')
        r = subprocess.run([cxx, '-O2', '-std=c++17', '-shared', '-fPIC', '-Wall', CPU_TWIN_SRC, '-o', CPU_TWIN_LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('the CPU twin failed to compile:\n%s' % r.stderr)
    return CPU_TWIN_LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
    print(build_cpu_twin(force='--force' in sys.argv))
