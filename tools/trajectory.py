"""Training trajectory of the HIP path next to the independent torch restatement (oracle/torch_ref.py) run THROUGH torch-ROCm on the
same GPU (test infrastructure: fp64 autograd of a few hundred steps takes hours on the CPU), same initial variables, same batches,
same TF-1.4 Adam: both loss curves step by step and how far they drift apart.

    python tools/trajectory.py [--enc av|a] [--batch 8] [--steps 250] [--pool 4] [--lr 1e-4] [--ref-dtype f64|f32] [--out FILE]

SAGEN_FP32_ONLY=1 in the environment runs the device side on the exact fp32 MFMA kernels instead of the bf16x3 ones."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spatialaudiogen_amd.model import SptAudioGen                     # noqa: E402
from spatialaudiogen_amd.weights import variable_specs, init_weights  # noqa: E402
from spatialaudiogen_amd.train import Trainer, synthetic_batches, learning_rate, ADAM_BETA1, ADAM_BETA2, ADAM_EPSILON   # noqa: E402
from oracle.torch_ref import TorchRef                                 # noqa: E402


def run(enc, B, steps, pool, lr, ref_dtype, seed=5, log=print, device_only=False):
    torch.cuda.set_device(0)
    P = init_weights(variable_specs(enc), seed=0, mode='bench')
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    tr = Trainer(net, batch=B, lr=lr, lr_iters=250000, lr_decay=0.5)
    it = synthetic_batches(enc, B, seed=seed, pool=pool)
    batches = []
    for _ in range(pool):
        a, v, f, t, m = next(it)
        batches.append([None if x is None else torch.as_tensor(x).cuda() for x in (a, v, f, t, m)])
    dt = torch.float64 if ref_dtype == 'f64' else torch.float32
    ref = None if device_only else TorchRef(P, enc, dtype=dt, device='cuda')
    names = [k for k in P if '/moving_' not in k]
    if ref is not None:
        m1 = {k: torch.zeros_like(ref.P[k]) for k in names}
        m2 = {k: torch.zeros_like(ref.P[k]) for k in names}
    b1, b2 = np.float32(ADAM_BETA1), np.float32(ADAM_BETA2)           # TF casts the betas to the variable dtype (oracle/np_oracle.py: adam_tf)
    omb1, omb2 = float(np.float32(1) - b1), float(np.float32(1) - b2)
    dev_l, ref_l = [], []
    t_dev = t_ref = 0.0
    for s in range(steps):
        a, v, f, t, m = batches[s % pool]
        t0 = time.time()
        loss, _ = tr.step(a, v, f, t, m)
        dev_l.append(float(loss))
        t_dev += time.time() - t0
        if ref is not None:
            t0 = time.time()
            l, g = ref.loss_and_grad_tensors(a, v, f, t, m[:, 1:])
            lr_s = learning_rate(s, lr, 250000, 0.5)
            lr_t = lr_s * np.sqrt(1 - ADAM_BETA2 ** (s + 1)) / (1 - ADAM_BETA1 ** (s + 1))
            for k in names:
                m1[k].mul_(float(b1)).add_(g[k], alpha=omb1)
                m2[k].mul_(float(b2)).addcmul_(g[k], g[k], value=omb2)
                ref.P[k] = ref.P[k] - lr_t * m1[k] / (m2[k].sqrt() + ADAM_EPSILON)
            ref_l.append(float(l))
            t_ref += time.time() - t0
            if s % 10 == 0 or s == steps - 1:
                log('step %4d  device %.6g  reference %.6g  rel.dev %.3g' % (s, dev_l[-1], ref_l[-1], abs(dev_l[-1] - ref_l[-1]) / max(abs(ref_l[-1]), 1e-30)))
        elif s % 50 == 0:
            log('step %4d  device %.6g' % (s, dev_l[-1]))
    return np.array(dev_l), np.array(ref_l), t_dev, t_ref


def summarise(dev_l, ref_l, log=print):
    rel = np.abs(dev_l - ref_l) / np.maximum(np.abs(ref_l), 1e-30)
    lr_ = np.abs(np.log(np.maximum(dev_l, 1e-30) / np.maximum(ref_l, 1e-30)))
    n = len(dev_l)
    rows = []
    for lo in range(0, n, max(1, n // 10)):
        hi = min(n, lo + max(1, n // 10))
        rows.append((lo, hi, float(np.median(rel[lo:hi])), float(rel[lo:hi].max()), float(np.mean(dev_l[lo:hi])), float(np.mean(ref_l[lo:hi]))))
        log('steps %4d-%4d  rel.dev median %.3g max %.3g   mean loss device %.6g reference %.6g' % rows[-1])
    log('whole run: rel.dev median %.3g, 90th pct %.3g, max %.3g; |log ratio| median %.3g max %.3g; final-10 mean device %.6g reference %.6g; '
        'first-10 mean %.6g' % (np.median(rel), np.percentile(rel, 90), rel.max(), np.median(lr_), lr_.max(), dev_l[-10:].mean(), ref_l[-10:].mean(), dev_l[:10].mean()))
    return rel


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--enc', default='av')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--steps', type=int, default=250)
    ap.add_argument('--pool', type=int, default=4)
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--ref-dtype', default='f64')
    ap.add_argument('--device-only', action='store_true')
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    enc = ['audio', 'video'] if args.enc == 'av' else ['audio']
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)
    log('trajectory: %s, B=%d, %d steps on %d repeated batches, lr %g, reference %s on torch-ROCm, device kernels: %s'
        % ('+'.join(enc), args.batch, args.steps, args.pool, args.lr, 'none' if args.device_only else args.ref_dtype,
           'exact fp32 MFMA (SAGEN_FP32_ONLY=1)' if os.environ.get('SAGEN_FP32_ONLY') else 'bf16x3'))
    d, r, td, trf = run(enc, args.batch, args.steps, args.pool, args.lr, args.ref_dtype, log=log, device_only=args.device_only)
    log('time: device %.2f s (%.2f ms/step incl. the loss read-back), reference %.2f s (%.1f ms/step)' % (td, 1e3 * td / args.steps, trf, 1e3 * trf / args.steps))
    if len(r):
        summarise(d, r, log)
    else:
        n = len(d)
        for lo in range(0, n, max(1, n // 30)):
            w = d[lo:lo + max(1, n // 30)]
            log('step %5d  loss mean %.6g  min %.6g  max %.6g' % (lo, w.mean(), w.min(), w.max()))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, 'w').write('\n'.join(lines) + '\n')
        np.savez(os.path.splitext(args.out)[0] + '.npz', device=d, reference=r)
