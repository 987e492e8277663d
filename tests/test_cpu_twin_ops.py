"""The op-level parity cases of tests/test_gpu_ops.py against the CPU twin of include/sagen.h's op level (libsagen_cpu.so,
spatialaudiogen_amd/csrc_cpu/sagen_cpu.cpp) - in THIS container, without a GPU.  The twin is selected explicitly
(SAGEN_LIB), implements nothing above the op level, and is never a fallback: the last test checks that."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ('test_stft_mag_and_spec or test_stft_known_answer_sinusoid or test_conv_2d or test_maxpool_bn_relu or test_fully_connected or '
         'test_deconv_2d or test_deconv_delta_weight_pins_offsets or test_mask_istft_mix or test_identity_mask_round_trip or test_power_map or '
         'test_assemble_wyzx_is_bit_exact')


@pytest.fixture(scope='module')
def twin():
    from spatialaudiogen_amd import build
    return build.build_cpu_twin()


def test_op_level_cases_pass_on_the_cpu_twin(twin):
    env = dict(os.environ, SAGEN_LIB=twin)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_ops.py'), '-m', 'gpu', '-q', '-x', '-k', CASES,
                        '-p', 'no:cacheprovider'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout, r.stdout[-500:]


def test_the_twin_is_op_level_only_and_never_a_fallback(twin):
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from spatialaudiogen_amd import _lib\n"
            "l = _lib.lib(); assert _lib.IS_CPU_TWIN and l.sagen_build_info().decode().startswith('cpu-twin')\n"
            "try:\n    l.sagen_create(None, None); raise SystemExit('the twin answered a context call')\n"
            "except _lib.SagenError as e:\n    assert 'op level' in str(e)\n" % ROOT)
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, SAGEN_LIB=twin), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    # without SAGEN_LIB the package never looks at the twin: the default library path is the HIP one
    from spatialaudiogen_amd import _lib
    assert os.path.basename(_lib.LIB_PATH) == 'libsagen_hip.so' or os.environ.get('SAGEN_LIB')
