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


def oracle_window(audio, row, size=52799):
    """One input window cut by the ORACLE's window table (oracle.np_oracle.deploy_window_table row: t, start_frame, pad_before,
    frame_idx, batch_id, read_start) - not by the product's deploy.audio_window: `pad_before` zeros, then the samples from
    `read_start`, zero-filled to `size` at the end of the clip.  audio [n, C]."""
    pad_before, read_start = int(row[2]), int(row[5])
    out = np.zeros((size, audio.shape[1]), audio.dtype)
    chunk = audio[read_start:read_start + size - pad_before]
    out[pad_before:pad_before + chunk.shape[0]] = chunk
    return out
