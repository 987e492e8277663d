"""Diagnostic: whole-network gradients under debugging switches (subprocess per setting), first mismatching variable and determinism."""
import os, sys, subprocess, tempfile, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))

CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from test_gpu_backward import _setup
from spatialaudiogen_amd.train import Trainer
enc = sys.argv[2].split('+'); B = int(sys.argv[3]); seed = int(sys.argv[4])
net, ref, P, inp, target = _setup(torch, enc, B, seed)
tr = Trainer(net, batch=B)
out = {}
for rep in range(2):
    tr.forward_backward(inp['audio'], inp.get('video'), inp.get('flow'), target, update_moving=False)
    torch.cuda.synchronize()
    for k in tr.opt.layout:
        out['%%d|%%s' %% (rep, k.replace('/', '|'))] = tr.grad(k).cpu().numpy()
    for b in ('t:g:feat', 't:g:fcred', 't:dy:vfc'):
        out['%%d|buf|%%s' %% (rep, b)] = tr.buffer(b).cpu().numpy()
np.savez(sys.argv[1], **out)
'''


def main():
    import torch
    from test_gpu_backward import _setup
    from util import rel_rms_err
    enc, B, seed = sys.argv[1].split('+'), int(sys.argv[2]), int(sys.argv[3])
    net, ref, P, inp, target = _setup(torch, enc, B, seed)
    keep = ('video_encoder/conv5_2',)
    loss, grads, pred, ig = ref.loss_and_grads(inp['audio'], inp.get('video'), inp.get('flow'), target, None, keep=keep)
    gfeat = np.transpose(ig['video_encoder/conv5_2'], (0, 2, 3, 1)).reshape(-1)
    settings = [{}, {'SAGEN_TRAIN_NO_H2W': '1'}, {'SAGEN_WGRAD_NO_H2': '1'}, {'SAGEN_TRAIN_NO_H2D': '1'}, {'SAGEN_ONE_STREAM': '1'}, {'SAGEN_BWD_NOSPLIT': '1'}, {'SAGEN_FP32_ONLY': '1'}, {'SAGEN_NO_P3': '1'}, {'SAGEN_WGRAD_REF': '1'}]
    if os.environ.get('DIAG_SETTINGS'):
        settings = settings[:int(os.environ['DIAG_SETTINGS'])]
    for st in settings:
        fn = tempfile.mktemp(suffix='.npz')
        env = dict(os.environ); env.update(st)
        r = subprocess.run([sys.executable, '-c', CHILD % (ROOT, os.path.join(ROOT, 'tests')), fn, '+'.join(enc), str(B), str(seed)], env=env,
                           capture_output=True, text=True)
        if r.returncode:
            print(st, 'FAILED', r.stderr[-2000:]); continue
        z = dict(np.load(fn))
        names = list(grads)
        errs = [(k, rel_rms_err(z['0|' + k.replace('/', '|')], grads[k])) for k in names]
        bad = [(k, e) for k, e in errs if e > 2e-4]
        same = all(np.array_equal(z['0|' + k.replace('/', '|')], z['1|' + k.replace('/', '|')]) for k in names)
        print('== %s: %d bad of %d; max err %.2e; run-to-run identical: %s; g:feat err %.2e (2nd run %.2e)' % (
            st, len(bad), len(errs), max(e for _, e in errs), same, rel_rms_err(z['0|buf|t:g:feat'], gfeat), rel_rms_err(z['1|buf|t:g:feat'], gfeat)))
        for k, e in bad[:16]:
            print('  bad  %-55s %.2e' % (k, e))
        order = [k for k in names if 'video_encoder' in k or 'video-fc' in k]
        for k, e in errs:
            if k in order[-14:] or k in order[:3]:
                print('     %-55s %.2e' % (k, e))


if __name__ == '__main__':
    main()
