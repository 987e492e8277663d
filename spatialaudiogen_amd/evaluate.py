"""Evaluation driver — the on-graph part of the reference's eval.py (eval.py:29-232) on the HIP path, sharded over
GPUs.

    torchrun --nproc-per-node 8 -m spatialaudiogen_amd.evaluate <model_dir> <db_dir> --subset_fn <list.txt>

Per clip: every 10th 0.1 s window (`skip_rate=10`, feeder.py:379), batches of 16 (eval.py:44), input = W channel with
1 s of context, target = Y,Z,X of the centre 0.1 s (eval.py:68-72), channel masks per clip (feeder.py:312-314,
`meta/audio_layouts.txt`).  Each rank owns a contiguous block of clips (whole batches stay on one GPU because
batch-norm runs on batch statistics); per-sample metric rows are written by rank 0 to `eval-detailed.txt` in the
reference's format, and the global means come from ONE all-reduce of float64 sums + count (RCCL).

Host-only metrics of eval.py (mel-LSD via librosa, envelope distance via scipy.hilbert, EMD via pyemd) are outside
this path and are not computed.
"""
import os
from collections import OrderedDict

import numpy as np

from .definitions import VIDEO, FLOW, NO_SEPARATION
from .dist import init_process_group, shard_range, MetricReducer

BATCH_SIZE = 16            # eval.py:44
METRIC_KEYS = ['amplitude/predicted', 'amplitude/gt',
               'mse/avg', 'mse/X', 'mse/Y', 'mse/Z', 'stft/avg', 'stft/X', 'stft/Y', 'stft/Z',
               'lsd/avg', 'lsd/X', 'lsd/Y', 'lsd/Z', 'snr/avg', 'snr/X', 'snr/Y', 'snr/Z']    # eval.py:125-133 (on-graph subset)


def read_layouts(fn):
    """meta/audio_layouts.txt: '<id> WXYZ|WXY' -> channel mask over (W, Y, Z, X) (feeder.py:312-314)."""
    masks = {'WXYZ': np.array([1., 1., 1., 1.]), 'WXY': np.array([1., 1., 0., 1.])}
    out = {}
    if fn and os.path.exists(fn):
        for l in open(fn).read().splitlines():
            if l.strip():
                out[l.split()[0]] = masks[l.split()[1]]
    return out


def clip_windows(reader):
    """All samples of one clip as stacked arrays."""
    chunks = list(reader.loop_chunks())
    if not chunks:
        return None
    out = {'id': [c['id'] for c in chunks], 'ambix': np.stack([c['ambix'] for c in chunks], 0).astype(np.float32)}
    for k in ('video', 'flow'):
        if k in chunks[0]:
            out[k] = np.stack([c[k] for c in chunks], 0).astype(np.float32)
    return out


class Evaluator(object):
    def __init__(self, net, params):
        self.net, self.params = net, params
        self.ss = int(params.audio_rate * params.context) // 2
        self.t = int(params.audio_rate * 0.1)
        self.rows, self.ids = [], []

    def run_batch(self, ids, ambix, video, flow, masks):
        """ambix [n<=16, 52799, 4]; masks [n, 4].  Short batches are zero-padded to 16 like the TF queue would never
        do (it blocks) — the padded windows are dropped from the metrics but do enter batch-norm statistics, so a
        clip list whose window count is not a multiple of 16 ends with one such batch per rank."""
        import torch
        n = ambix.shape[0]
        pad = lambda x: x if x is None or n == BATCH_SIZE else np.concatenate([x, np.zeros((BATCH_SIZE - n,) + x.shape[1:], x.dtype)], 0)
        dev = self.net.device
        a = torch.as_tensor(pad(ambix)).to(dev)
        pred = self.net.inference_ops(a[:, :, :1].contiguous(), pad(video), pad(flow))
        target = a[:, self.ss:self.ss + self.t, 1:].contiguous()
        m = torch.as_tensor(pad(masks.astype(np.float32))).to(dev)
        _, stft_ps, lsd_ps, mse_ps, snr_ps = self.net.evaluation_ops(pred, target, None, m[:, 1:])
        amp_p = pred.abs().amax(dim=(1, 2)); amp_g = target.abs().amax(dim=(1, 2))
        per = torch.stack([amp_p, amp_g], 1).cpu().numpy()
        S = [x.cpu().numpy() for x in (mse_ps, stft_ps, lsd_ps, snr_ps)]
        for i in range(n):
            row = [per[i, 0], per[i, 1]]
            for arr in S:                       # eval.py:155-171 appends the RAW per-sample values (no x5e3 / x100)
                v = arr[i]
                row += [float(np.mean(v)), float(v[2]), float(v[0]), float(v[1])]   # avg, X, Y, Z  (channels are Y,Z,X)
            self.rows.append(row)
            self.ids.append(ids[i])


def evaluate(model_dir, db_dir, subset_fn=None, layouts_fn=None, variables=None, params=None, overwrite=True):
    import torch
    from .deploy import load_params, W2XYZ
    from .feeder import SampleReader, img_prep_fcn
    rank, world = init_process_group()
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    params = params or load_params(model_dir)
    w2 = W2XYZ(model_dir, params=params, variables=variables)
    net = w2.model
    ids = [l.strip() for l in open(subset_fn)] if subset_fn else sorted(os.listdir(db_dir))
    ids = [i for i in ids if i and os.path.isdir(os.path.join(db_dir, i))]
    layouts = read_layouts(layouts_fn)
    lo, hi = shard_range(len(ids), rank, world)
    ev = Evaluator(net, params)
    pend = {'id': [], 'ambix': [], 'video': [], 'flow': [], 'mask': []}

    def flush(force=False):
        while len(pend['id']) >= BATCH_SIZE or (force and pend['id']):
            take = min(BATCH_SIZE, len(pend['id']))
            cut = {k: v[:take] for k, v in pend.items()}
            for k in pend:
                pend[k] = pend[k][take:]
            stack = lambda k: np.stack(cut[k], 0) if cut[k] else None
            ev.run_batch(cut['id'], stack('ambix'), stack('video'), stack('flow'), stack('mask'))

    for yid in ids[lo:hi]:
        rd = SampleReader(os.path.join(db_dir, yid), ambi_order=params.ambi_order, audio_rate=params.audio_rate,
                          video_rate=params.video_rate, context=params.context, duration=0.1,
                          return_video=VIDEO in params.encoders, img_prep=img_prep_fcn(),
                          return_flow=FLOW in params.encoders, skip_silence_thr=None, shuffle=False,
                          random_rotations=False, skip_rate=10)                    # feeder.py:373-396 (for_eval)
        win = clip_windows(rd)
        if win is None:
            continue
        mask = layouts.get(yid, np.ones(4))
        for i in range(len(win['id'])):
            pend['id'].append(win['id'][i]); pend['ambix'].append(win['ambix'][i]); pend['mask'].append(mask)
            for k in ('video', 'flow'):
                if k in win:
                    pend[k].append(win[k][i])
        flush()
    flush(force=True)

    red = MetricReducer(METRIC_KEYS, device=net.device if world > 1 and torch.cuda.is_available() else None)
    rows = np.asarray(ev.rows, np.float64).reshape(-1, len(METRIC_KEYS))
    for r in rows:
        red.add(np.nan_to_num(r), 1)
    means, count = red.reduce()
    # per-sample rows to rank 0 (eval.py:210-215 file format)
    all_ids, all_rows = [ev.ids], [rows]
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, (ev.ids, rows))
        all_ids, all_rows = [g[0] for g in gathered], [g[1] for g in gathered]
    if rank == 0:
        eval_fn = os.path.join(model_dir, 'eval-detailed.txt')
        assert overwrite or not os.path.exists(eval_fn), 'Evaluation file already exists.'
        with open(eval_fn, 'w') as f:
            f.write('SampleID | {}\n'.format(' '.join(METRIC_KEYS)))
            for ids_r, rows_r in zip(all_ids, all_rows):
                for sid, row in zip(ids_r, rows_r):
                    f.write('{} | {}\n'.format(sid, ' '.join(str(v) for v in row)))
    return OrderedDict((k, means[k]) for k in METRIC_KEYS), count


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('model_dir')
    ap.add_argument('db_dir', help='directory of preprocessed clip folders (params.db_dir in the reference)')
    ap.add_argument('--subset_fn', default=None)
    ap.add_argument('--layouts_fn', default='meta/audio_layouts.txt')
    ap.add_argument('--overwrite', action='store_true')
    args = ap.parse_args(argv)
    means, count = evaluate(args.model_dir, args.db_dir, args.subset_fn, args.layouts_fn, overwrite=args.overwrite)
    if int(os.environ.get('RANK', 0)) == 0:
        print('EVAL | %d samples' % count)
        for k, v in means.items():
            print('EVAL | \t %s %f' % (k, v))


if __name__ == '__main__':
    main()
