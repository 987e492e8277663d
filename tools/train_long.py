"""BASELINE configs[4] as the reference states it (README.md:102, train.py:34-38, 205): audio+video, batch 32, Adam at lr 1e-4 with the
staircase decay 0.5 / 250 000 iterations, 150 000 iterations - on a NON-REPEATING synthetic stream generated on the device (the
statistics of weights.synth_inputs: three sinusoids + noise, 8x8-smoothed uint8 frames), target = the fixed per-channel mixing
of the mono crop that train.synthetic_batches uses.  Logs the loss (mean / min / max per window of iterations, read back once per
window), the step time and the NaN count.

    python tools/train_long.py [--iters 150000] [--log-every 1000] [--out gpurun_out/r04_train_150k.txt] [--enc av|a]"""
import argparse
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spatialaudiogen_amd.model import SptAudioGen                     # noqa: E402
from spatialaudiogen_amd.weights import variable_specs, init_weights  # noqa: E402
from spatialaudiogen_amd.train import Trainer                          # noqa: E402


class DeviceStream(object):
    """Endless stream of fresh synthetic batches, generated with torch on the GPU (no host work per step)."""

    def __init__(self, batch, video, seed=0):
        self.B, self.video = batch, video
        self.g = torch.Generator(device='cuda')
        self.g.manual_seed(seed)
        self.n = torch.arange(52799, device='cuda', dtype=torch.float32)[None, :]
        self.mix = torch.tensor([0.5, 0.25, -0.5], device='cuda')

    def next(self):
        B, g = self.B, self.g
        u = lambda *s: torch.rand(*s, device='cuda', generator=g)
        audio = torch.zeros(B, 52799, device='cuda')
        for _ in range(3):
            f = 100. + 7900. * u(B, 1)
            ph = 2 * math.pi * u(B, 1)
            audio += torch.sin(2 * math.pi * f * self.n / 48000. + ph)
        audio = (0.25 * audio + 0.05 * torch.randn(B, 52799, device='cuda', generator=g)).clamp_(-1, 1)[:, :, None].contiguous()
        video = None
        if self.video:
            img = torch.randint(0, 256, (B, 224, 448, 3), device='cuda', generator=g).float()
            sm = img.reshape(B, 28, 8, 56, 8, 3).mean(dim=(2, 4), keepdim=True).expand(B, 28, 8, 56, 8, 3).reshape(B, 224, 448, 3)
            video = (torch.round(0.5 * img + 0.5 * sm) / 255. - 0.5)[:, None].contiguous()
        target = (audio[:, 24000:28800, :] * self.mix).contiguous()
        return audio, video, None, target


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=150000)
    ap.add_argument('--log-every', type=int, default=1000)
    ap.add_argument('--enc', default='av')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--out', default='gpurun_out/r04_train_150k.txt')
    args = ap.parse_args()
    enc = ['audio', 'video'] if args.enc == 'av' else ['audio']
    torch.cuda.set_device(0)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(init_weights(variable_specs(enc), seed=0, mode='bench', fc3_std=0.001))   # the reference's initialisers (model.py:255)
    tr = Trainer(net, batch=args.batch, lr=1e-4, lr_iters=250000, lr_decay=0.5)                  # train.py:35-37
    a, v, f, t = DeviceStream(args.batch, 'video' in enc, seed=1).next()
    tr.autotune(a, v, f, t)
    stream = DeviceStream(args.batch, 'video' in enc, seed=2)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    out = open(args.out, 'w')

    def log(s):
        print(s, flush=True)
        out.write(s + '\n')
        out.flush()
    log('configs[4]: %s, batch %d, Adam lr 1e-4 (x0.5 every 250000), %d iterations, non-repeating synthetic stream generated on the device, '
        'kernels: %s' % ('+'.join(enc), args.batch, args.iters, 'exact fp32 MFMA' if os.environ.get('SAGEN_FP32_ONLY') else
                       'bf16x3' if (os.environ.get('SAGEN_NO_H2') or os.environ.get('SAGEN_TRAIN_NO_H2')) else
                       'fp16x2 planes for the stride-1 3x3 trunk convs (forward, data and weight gradients), bf16x3 elsewhere'))
    W = args.log_every
    buf = torch.zeros(W, dtype=torch.float64, device='cuda')
    nan_total = 0
    t0 = t_win = time.time()
    for it in range(args.iters):
        a, v, f, t = stream.next()
        loss, lr = tr.step(a, v, f, t)
        buf[it % W] = loss
        if (it + 1) % W == 0 or it + 1 == args.iters:
            n = (it % W) + 1
            w = buf[:n].cpu()                                                  # the only synchronisation of the window
            nans = int(torch.isnan(w).sum())
            nan_total += nans
            now = time.time()
            log('iter %6d  loss mean %.6g  min %.6g  max %.6g  lr %.3g  %.3f ms/iter  NaN %d' % (it + 1, float(w.nanmean()), float(w[~torch.isnan(w)].min()) if nans < n else float('nan'),
                float(w[~torch.isnan(w)].max()) if nans < n else float('nan'), lr, 1e3 * (now - t_win) / n, nans))
            t_win = now
            if nans:
                log('NaN loss: stopping (train.py:212-213 aborts the run)')
                break
    torch.cuda.synchronize()
    dt = time.time() - t0
    try:
        log('fp16x2 plane saturations over the whole run (elements clamped to +-65000): %d' % net.counter(args.batch, 'fp16x2_saturations'))
    except Exception as e:                                                     # (a build without the counter)
        log('fp16x2 saturation counter unavailable: %s' % e)
    log('done: %d iterations in %.1f s = %.3f ms/iteration incl. the batch generator = %.1f ambisonic-s/s trained; NaN count %d'
        % (it + 1, dt, 1e3 * dt / (it + 1), 0.1 * args.batch * (it + 1) / dt, nan_total))
