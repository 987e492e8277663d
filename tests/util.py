import numpy as np


def rms(x):
    return float(np.sqrt(np.mean(np.square(np.asarray(x, dtype=np.float64)))))


def rel_rms_err(got, ref):
    """RMS error relative to the RMS of the reference."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return rms(got - ref) / max(rms(ref), 1e-30)


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def ensure_lib():
    """GPU tests must run the HIP library; build it if the tree was shipped without the .so."""
    from spatialaudiogen_amd import _lib, build
    import os
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    l = _lib.lib()
    if not os.environ.get('SAGEN_LIB'):
        got, want = l.sagen_source_digest().decode(), build.source_digest()
        assert got == want, 'libsagen_hip.so was built from other sources (digest %s, tree %s): rebuild it' % (got, want)
    return l
