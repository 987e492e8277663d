"""Step 1 of the TF1 pin kit (runs anywhere this repository runs; numpy only): write the weights and inputs of the three golden
cases of tools/make_golden.py to ONE .npz that a Python 2 / numpy 1.14 / TensorFlow 1.4 machine can read (the seeds use numpy's
PCG64, which that numpy does not have - hence values, not seeds).

    python tools/tf1_pin_inputs.py [tf1_pin_inputs.npz]          # ~370 MB, float32; not committed

Keys: '<case>/var/<TF variable name>' (the checkpoint names of SURVEY.md 9.1) and '<case>/in/{audio,video,flow}'.
Step 2: tools/tf1_dump_golden.py on the TF1 machine.  Nothing of the reference is read here."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import make_golden as G
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs


def build():
    data = {}
    for name, (enc, B, ws, ins) in G.CASES.items():
        P = init_weights(variable_specs(enc), seed=ws, mode='test')
        inp = synth_inputs(B, enc, seed=ins)
        for k, v in P.items():
            data['%s/var/%s' % (name, k)] = np.asarray(v, np.float32)
        for k, v in inp.items():
            data['%s/in/%s' % (name, k)] = np.asarray(v, np.float32)
        data['%s/encoders' % name] = np.array(','.join(enc))
    return data


if __name__ == '__main__':
    fn = sys.argv[1] if len(sys.argv) > 1 else 'tf1_pin_inputs.npz'
    d = build()
    np.savez(fn, **d)
    print('wrote %s: %d arrays, %.0f MB' % (fn, len(d), sum(v.nbytes for v in d.values()) / 1e6))
