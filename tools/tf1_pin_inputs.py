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


def fingerprint(P, inp):
    """What ties an output file to the EXACT weights and inputs it was computed from: float64 [number of variables, sum over the
    variables of sum(v), of sum(v^2), then sum / sum of squares of every input present].  tools/tf1_dump_golden.py copies it through
    ('<case>/fingerprint'); tests/test_golden.py recomputes it from the current case table and FAILS on a mismatch (a stale pin)."""
    f = [float(len(P)), sum(float(np.asarray(v, np.float64).sum()) for v in P.values()),
         sum(float((np.asarray(v, np.float64) ** 2).sum()) for v in P.values())]
    for k in ('audio', 'video', 'flow'):
        if k in inp:
            x = np.asarray(inp[k], np.float32).astype(np.float64)
            f += [float(x.sum()), float((x ** 2).sum())]
    return np.array(f, np.float64)


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
        data['%s/fingerprint' % name] = fingerprint(P, inp)
    return data


if __name__ == '__main__':
    fn = sys.argv[1] if len(sys.argv) > 1 else 'tf1_pin_inputs.npz'
    d = build()
    np.savez(fn, **d)
    print('wrote %s: %d arrays, %.0f MB' % (fn, len(d), sum(v.nbytes for v in d.values()) / 1e6))
