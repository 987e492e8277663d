"""The C-ABI library builds for gfx950 without a GPU, loads, exports every symbol include/sagen.h
declares, and its host-side logic (configuration checks, variable inventory, workspace sizing, error
reporting) works without a device.  No compute entry point is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from spatialaudiogen_amd import build, _lib
    build.build(verbose=False)
    return _lib.lib()


def declared_functions():
    text = open(os.path.join(ROOT, 'include', 'sagen.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sagen_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from spatialaudiogen_amd import _lib
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), 'libsagen_hip.so does not export %s' % n
    assert sorted(_lib.SIGNATURES) == names, 'ctypes signatures out of sync with include/sagen.h'
    assert lib.sagen_version() == 100


def _cfg(**kw):
    from spatialaudiogen_amd._lib import SagenConfig
    c = SagenConfig()
    c.batch, c.encoders, c.separation, c.num_sep_tracks, c.n_loc_units = 4, 3, 1, 32, 2
    c.loc_units[0], c.loc_units[1] = 512, 512
    c.ambi_order, c.audio_rate, c.video_rate = 1, 48000, 10
    c.context, c.sample_duration, c.fft_window = 1.0, 0.1, 0.025
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_context_inventory_matches_python_inventory(lib):
    from spatialaudiogen_amd.weights import variable_specs
    for enc_mask, enc in ((1, ['audio']), (3, ['audio', 'video']), (7, ['audio', 'video', 'flow'])):
        h = C.c_void_p()
        cfg = _cfg(encoders=enc_mask)
        assert lib.sagen_create(C.byref(h), C.byref(cfg)) == 0
        specs = variable_specs(enc)
        n = lib.sagen_num_variables(h)
        assert n == len(specs)
        name, ndim, shape = C.c_char_p(), C.c_int32(), (C.c_int64 * 4)()
        got = {}
        for i in range(n):
            assert lib.sagen_variable_spec(h, i, C.byref(name), C.byref(ndim), shape) == 0
            got[name.value.decode()] = tuple(shape[k] for k in range(ndim.value))
        assert got == {k: tuple(v) for k, v in specs.items()}
        ws = lib.sagen_workspace_bytes(h)
        assert ws > 4 * sum(int(np.prod(v)) for v in specs.values())      # packed filters + activations
        lib.sagen_destroy(h)


def test_bad_configurations_fail_with_codes_and_messages(lib):
    h = C.c_void_p()
    assert lib.sagen_create(C.byref(h), None) == -1                                    # SAGEN_ERR_NULL
    assert lib.sagen_create(C.byref(h), C.byref(_cfg(encoders=2))) == -3               # audio is mandatory
    assert b'audio encoder' in lib.sagen_last_error()
    assert lib.sagen_create(C.byref(h), C.byref(_cfg(audio_rate=44100))) == -3         # SAGEN_ERR_UNSUPPORTED
    assert lib.sagen_create(C.byref(h), C.byref(_cfg(batch=0))) == -2                  # SAGEN_ERR_SHAPE
    assert lib.sagen_create(C.byref(h), C.byref(_cfg(num_sep_tracks=7))) == -3
    assert lib.sagen_create(C.byref(h), C.byref(_cfg())) == 0
    assert lib.sagen_forward(h, None, None, None, None, None) == -1
    buf = (C.c_float * 4)()
    assert lib.sagen_forward(h, buf, buf, None, buf, None) == -4                       # weights not bound
    assert lib.sagen_bind_weights(h, None, 0, None, 0, None) == -1
    lib.sagen_destroy(h)
    assert lib.sagen_stft_mag(None, 1, 52799, 0, 200, None, 0, 0, None, None) == -1
    assert lib.sagen_conv2d_scratch_bytes(2, 8, 8, 3, 3, 64, 64) >= 64 * 576 * 4


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under spatialaudiogen_amd/ may reference it."""
    pkg = os.path.join(ROOT, 'spatialaudiogen_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.lower().replace('np_oracle', 'oracle') or f == 'never', (f, 'mentions the oracle')


def test_tile_table_is_exposed_and_consistent():
    """sagen_num_tiles / sagen_tile_name (host-only): the table SptAudioGen.tile_names, plan files and forced plans index."""
    from spatialaudiogen_amd import _lib
    from spatialaudiogen_amd.model import SptAudioGen
    L = _lib.lib()
    n = L.sagen_num_tiles()
    names = [L.sagen_tile_name(i).decode() for i in range(n)]
    assert n >= 40 and len(set(names)) == n
    assert L.sagen_tile_name(-1) is None and L.sagen_tile_name(n) is None
    assert names == SptAudioGen.tile_names()
    fams = {nm.split('<')[0] for nm in names}
    assert fams == {'igemm_kernel', 'igemm3_kernel', 'igemm3dw_kernel', 'igemm3s2_kernel', 'conv3p_kernel', 'conv3pp_kernel', 'conv3g_kernel', 'conv3h_kernel', 'conv3hr_kernel'}
