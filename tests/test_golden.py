"""Committed golden fixtures (tests/golden/golden_v1.npz, made by tools/make_golden.py from the fp64
oracle; inputs/weights are regenerated from seeds).  CPU: the oracle still reproduces them (pins the
restatement against silent edits).  GPU: the HIP path reproduces them to the parity bar."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import make_golden as G          # noqa: E402  (case table + probe list)
from util import rms             # noqa: E402

GOLD = np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_v1.npz'))


@pytest.mark.parametrize('name', ['a_b2_s0', 'av_b2_s1'])
def test_oracle_reproduces_golden(name):
    out = G.run_case(name)
    for k, v in out.items():
        ref = GOLD[k]
        if k.endswith('/ambix'):
            assert rms(v.astype(np.float64) - ref) < 1e-6 * max(rms(ref), 1e-9), k
        else:
            assert np.allclose(v, ref, rtol=1e-7, atol=1e-9), k


def test_deploy_table_fixture():
    from oracle import np_oracle as O
    rows = np.array(O.deploy_window_table(O.audio_pow_times(12), 0., 10.), dtype=np.float64)
    assert np.array_equal(rows, GOLD['deploy_table_12s'])


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(G.CASES))
def test_hip_path_reproduces_golden(name):
    import torch
    from spatialaudiogen_amd.model import SptAudioGen
    from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
    assert torch.cuda.is_available()
    enc, B, ws, ins = G.CASES[name]
    P = init_weights(variable_specs(enc), seed=ws, mode='test')
    inp = synth_inputs(B, enc, seed=ins)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    got = net.inference_ops(inp['audio'], inp.get('video'), inp.get('flow')).cpu().numpy()
    ref = GOLD[name + '/ambix']
    err = rms(got - ref)
    assert err <= 1e-4 and err <= 1e-3 * rms(ref), (name, err, rms(ref))
