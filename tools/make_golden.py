"""Regenerate tests/golden/*.npz from the CPU oracle (fp64).  The reference itself cannot run here
(TF1 / Python 2), so these are regression fixtures of the restatement: outputs and
checksums only — inputs and weights are regenerated from numpy PCG64 seeds, never stored.

    python tools/make_golden.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import np_oracle as O
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs

CASES = {   # name: (encoders, batch, weight seed, input seed)
    'a_b2_s0': (['audio'], 2, 0, 1234),
    'av_b2_s1': (['audio', 'video'], 2, 1, 1235),
    'avf_b1_s2': (['audio', 'video', 'flow'], 1, 2, 1236),
}
PROBES = ['audio_encoder/mag', 'audio_encoder/conv1', 'audio_encoder/conv5', 'video_encoder/conv5_2',
          'flow_encoder/conv5_2', 'bottleneck', 'localization/coeffs', 'separation/deconv1', 'decoder/ambix']


def run_case(name):
    enc, B, ws, ins = CASES[name]
    P = init_weights(variable_specs(enc), seed=ws, mode='test')
    inp = synth_inputs(B, enc, seed=ins)
    orc = O.SptAudioGenOracle(encoders=enc)
    y = orc.inference_ops(inp['audio'], P, video=inp.get('video'), flow=inp.get('flow'))
    out = {name + '/ambix': y.astype(np.float32)}
    for p in PROBES:
        if p in orc.ends:
            v = np.asarray(orc.ends[p], np.float64)
            out['%s/chk/%s' % (name, p)] = np.array([v.sum(), (v ** 2).sum()] + list(v.reshape(-1)[:: max(v.size // 16, 1)][:16]))
    return out


if __name__ == '__main__':
    data = {}
    for name in CASES:
        data.update(run_case(name))
        print('done', name)
    rows = O.deploy_window_table(O.audio_pow_times(12), 0., 10.)
    data['deploy_table_12s'] = np.array(rows, dtype=np.float64)
    os.makedirs(os.path.join(ROOT, 'tests', 'golden'), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'golden_v1.npz'), **data)
    print('wrote', sum(v.nbytes for v in data.values()), 'bytes')
