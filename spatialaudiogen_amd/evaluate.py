"""Evaluation driver — the on-graph part of the reference's eval.py (eval.py:29-232) on the HIP path, sharded over
GPUs.

    torchrun --nproc-per-node 8 -m spatialaudiogen_amd.evaluate <model_dir> <db_dir> --subset_fn <list.txt>

Per clip: every 10th 0.1 s window (`skip_rate=10`, feeder.py:379), batches of 16 (eval.py:44), input = W channel with
1 s of context, target = Y,Z,X of the centre 0.1 s (eval.py:68-72), channel masks per clip (feeder.py:312-314,
`meta/audio_layouts.txt`).  Each rank owns a contiguous block of clips (whole batches stay on one GPU because
batch-norm runs on batch statistics; batches are cut from ONE global window order, so results do not depend on the number
of ranks); per-sample metric rows are written by rank 0 to `eval-detailed.txt` in the
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
SKIP_RATE = 10             # feeder.py:379 (for_eval)
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


def window_plan(db_dir, clip_ids):
    """The GLOBAL evaluation order: clips in list order, within a clip every SKIP_RATE-th window of audio_pow.lst in time
    order.  Only the (small) lists are read.  Returns [(clip_id, t)].

    The reference fills one TF queue from 4 racing reader threads (feeder.py:372-410) and dequeues 16 at a time, so its
    batch composition is not reproducible; because batch-norm runs on batch statistics the composition is part of the
    result, and this driver therefore fixes it: batch b = windows [16 b, 16 b + 16) of this order - for ANY number of ranks."""
    from .feeder import read_pow_list, select_times
    plan = []
    for yid in clip_ids:
        times, powers = read_pow_list(os.path.join(db_dir, yid, 'audio_pow.lst'))
        plan.extend((yid, t) for t in select_times(times, powers, skip_rate=SKIP_RATE))
    return plan


def batch_shard(n_windows, rank, world, partial_batch='drop'):
    """Whole batches per rank: (first_batch, end_batch, n_batches).  The trailing partial batch is dropped like the
    reference's queue drops it (dequeue_many blocks, eval.py:140-143 / feeder.py:412-419 stop with up to 31 samples
    queued) or, with partial_batch='pad', evaluated zero-padded as the LAST batch."""
    n_batches = n_windows // BATCH_SIZE + (1 if partial_batch == 'pad' and n_windows % BATCH_SIZE else 0)
    lo, hi = shard_range(n_batches, rank, world)
    return lo, hi, n_batches


class Evaluator(object):
    def __init__(self, net, params, power_maps=False):
        self.net, self.params = net, params
        self.ss = int(params.audio_rate * params.context) // 2
        self.t = int(params.audio_rate * 0.1)
        self.rows, self.ids = [], []
        self.power_maps = power_maps
        self.maps = []                    # per sample (pred map, gt map) [7,12] each (eval.py:190: ang_res=30)
        self._sh = None

    def run_batches(self, batches, grouped_net):
        """`len(batches)` FULL batches (each (ids, ambix [16,52799,4], video, flow, masks), frames of one dtype) as ONE grouped forward
        call (round 6: SptAudioGen(groups=G): a launch per layer for all of them, every batch with its own batch-norm statistics - the
        predictions are bit-identical to run_batch's) followed by ONE metrics launch over all their windows (per-sample values)."""
        import torch
        dev = self.net.device
        cat = lambda k: None if batches[0][k] is None else torch.cat([torch.as_tensor(b[k]) for b in batches], 0).to(dev)
        a = cat(1)
        pred = grouped_net.inference_ops_checked(a[:, :, :1].contiguous(), cat(2), cat(3), on_saturation='raise')
        m = torch.as_tensor(np.concatenate([b[4].astype(np.float32) for b in batches], 0)).to(dev)
        self._finish([i for b in batches for i in b[0]], a, pred, m, a.shape[0])

    def run_batch(self, ids, ambix, video, flow, masks):
        """ambix [n<=16, 52799, 4]; masks [n, 4].  n < 16 only for the optional zero-padded final batch: its padded
        windows are dropped from the metrics but do enter the batch-norm statistics."""
        import torch
        n = ambix.shape[0]
        from .feeder import frames_to_float
        if n != BATCH_SIZE and video is not None:
            video = frames_to_float(video)              # the padding windows are 0.0 AFTER normalisation: no uint8 pixel maps to it
        pad = lambda x: x if x is None or n == BATCH_SIZE else np.concatenate([x, np.zeros((BATCH_SIZE - n,) + x.shape[1:], x.dtype)], 0)
        dev = self.net.device
        a = torch.as_tensor(pad(ambix)).to(dev)
        # (fp16x2 guard: a batch whose trunk planes clamped anything is re-run on bf16 planes - SptAudioGen.inference_ops_checked)
        pred = self.net.inference_ops_checked(a[:, :, :1].contiguous(), pad(video), pad(flow))
        m = torch.as_tensor(pad(masks.astype(np.float32))).to(dev)
        self._finish(ids, a, pred, m, n)

    def _finish(self, ids, a, pred, m, n):
        """metrics + rows of the first n windows of a forward's worth of windows (a [N,52799,4] on the device, pred [N,4800,3], m [N,4])"""
        import torch
        target = a[:, self.ss:self.ss + self.t, 1:].contiguous()
        _, stft_ps, lsd_ps, mse_ps, snr_ps = self.net.evaluation_ops(pred, target, None, m[:, 1:])
        amp_p = pred.abs().amax(dim=(1, 2)); amp_g = target.abs().amax(dim=(1, 2))
        per = torch.stack([amp_p, amp_g], 1).cpu().numpy()
        S = [x.cpu().numpy() for x in (mse_ps, stft_ps, lsd_ps, snr_ps)]
        if self.power_maps:
            self._power_maps(a, pred, target, m, n)
        for i in range(n):
            row = [per[i, 0], per[i, 1]]
            for k, arr in enumerate(S):         # eval.py:155-171 appends the RAW per-sample values (no x5e3 / x100)
                v = arr[i]
                avg = float(np.nanmean(v)) if k == 3 and np.isfinite(v).any() else float(np.mean(v))   # snr/avg is a nanmean (eval.py:168)
                row += [avg, float(v[2]), float(v[0]), float(v[1])]   # avg, X, Y, Z  (channels are Y,Z,X)
            self.rows.append(row)
            self.ids.append(ids[i])

    def _power_maps(self, a, pred, target, m, n):
        """Directional RMS maps of the masked WYZX prediction / ground truth of every sample - the inputs of the
        reference's EMD metric (eval.py:147-149,188-191 -> distance.py:41-59,133-143; one 0.1 s frame per sample, 30 degree
        mesh), computed on the device by sagen_power_map_batched.  The EMD itself (pyemd) is host code outside the path."""
        import torch
        from . import ops
        from .ambisonics import sh_matrix
        if self._sh is None:
            self._sh = torch.as_tensor(sh_matrix(30.0), dtype=torch.float32, device=pred.device)
        mono = a[:, self.ss:self.ss + self.t, :1]
        for x in (pred, target):
            wyzx = (torch.cat([mono, x], 2) * m[:, None, :]).contiguous()
            self.maps.append(ops.power_map_batched(wyzx, self._sh)[:n].reshape(n, 7, 12).flip(1).cpu().numpy())


class _ReaderCache(object):
    """SampleReaders of the most recently used clips (consecutive windows usually come from the same clip)."""

    def __init__(self, make, keep=2):
        self.make, self.keep, self.items = make, keep, OrderedDict()

    def get(self, yid):
        if yid in self.items:
            self.items.move_to_end(yid)
        else:
            self.items[yid] = self.make(yid)
            while len(self.items) > self.keep:
                self.items.popitem(last=False)
        return self.items[yid]


def evaluate(model_dir, db_dir, subset_fn=None, layouts_fn=None, variables=None, params=None, overwrite=True,
             partial_batch='drop', power_maps=False, groups=1):
    import torch
    from .deploy import load_params, W2XYZ
    from .feeder import SampleReader
    rank, world = init_process_group()
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
    params = params or load_params(model_dir)
    w2 = W2XYZ(model_dir, params=params, variables=variables)
    net = w2.model
    ids = [l.strip() for l in open(subset_fn)] if subset_fn else sorted(os.listdir(db_dir))
    ids = [i for i in ids if i and os.path.isdir(os.path.join(db_dir, i))]
    layouts = read_layouts(layouts_fn)
    plan = window_plan(db_dir, ids)
    lo, hi, n_batches = batch_shard(len(plan), rank, world, partial_batch)
    readers = _ReaderCache(lambda yid: SampleReader(
        os.path.join(db_dir, yid), ambi_order=params.ambi_order, audio_rate=params.audio_rate, video_rate=params.video_rate,
        context=params.context, duration=0.1, return_video=VIDEO in params.encoders, img_prep=None,      # frames stay uint8 (sagen_forward_u8)
        return_flow=FLOW in params.encoders, skip_silence_thr=None, shuffle=False, random_rotations=False,
        skip_rate=SKIP_RATE))                                              # feeder.py:373-396 (for_eval)
    ev = Evaluator(net, params, power_maps=power_maps)
    # groups > 1 (round 6): `groups` consecutive FULL batches of the rank's shard per forward call (W2XYZ._grouped_model: the same
    # device variables, grouped native contexts) - per-sample rows unchanged; a zero-padded partial batch and the shard's tail run singly
    w2.groups = max(1, int(groups))
    gnet = w2._grouped_model(w2.groups) if w2.groups > 1 else None
    pending = []

    def flush(force_single=False):
        if pending and (force_single or len(pending) < w2.groups):
            for q in pending:
                ev.run_batch(*q)
        elif pending:
            ev.run_batches(pending, gnet)
        del pending[:]
    for b in range(lo, hi):
        wins = plan[b * BATCH_SIZE:(b + 1) * BATCH_SIZE]
        samples = [readers.get(yid).sample_at(t) for yid, t in wins]
        def stack(k):
            if k not in samples[0]:
                return None
            x = np.stack([smp[k] for smp in samples], 0)
            return x if (k == 'video' and x.dtype == np.uint8) else x.astype(np.float32)
        item = ([smp['id'] for smp in samples], stack('ambix'), stack('video'), stack('flow'),
                np.stack([layouts.get(yid, np.ones(4)) for yid, _ in wins], 0))
        full = len(samples) == BATCH_SIZE and (item[2] is None or item[2].dtype == np.uint8)
        if gnet is None or not full:
            flush(force_single=True)
            ev.run_batch(*item)
            continue
        pending.append(item)
        if len(pending) == w2.groups:
            flush()
    flush(force_single=True)

    # global means (np.mean over every sample, eval.py:223): ONE all-reduce of per-key sums (+ finite-only sums / counts) + sample count
    red = MetricReducer(METRIC_KEYS, device=net.device if world > 1 and torch.cuda.is_available() and
                        torch.distributed.get_backend() != 'gloo' else None)
    rows = np.asarray(ev.rows, np.float64).reshape(-1, len(METRIC_KEYS))
    red.add_rows(rows)
    means, count = red.reduce()
    # per-sample rows to rank 0 (eval.py:210-215 file format); ranks hold consecutive batch ranges, so rank order = global order
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
        short = {k: c for k, c in red.finite_counts.items() if c < count}
        if short:        # the means above are np.mean over all samples (eval.py:223): a non-finite sample shows there; say how many
            print('EVAL | non-finite per-sample values: ' + ', '.join('%s %d/%d finite (finite-only mean %.6g)' % (k, c, count, red.finite_means[k])
                                                                     for k, c in sorted(short.items())))
        ev_sat = getattr(net, 'saturation_events', [])
        if ev_sat:
            print('EVAL | fp16x2 guard: %d batches clamped activation-plane elements (%d in all) and were re-run on bf16 planes'
                  % (len(ev_sat), sum(c for _, c in ev_sat)))
        dropped = len(plan) - count
        if dropped:
            print('EVAL | %d trailing windows (a partial batch of %d) were not evaluated' % (dropped, BATCH_SIZE))
    evaluate.last_maps = ev.maps
    return OrderedDict((k, means[k]) for k in METRIC_KEYS), count


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('model_dir')
    ap.add_argument('db_dir', help='directory of preprocessed clip folders (params.db_dir in the reference)')
    ap.add_argument('--subset_fn', default=None)
    ap.add_argument('--layouts_fn', default='meta/audio_layouts.txt')
    ap.add_argument('--overwrite', action='store_true')
    ap.add_argument('--groups', type=int, default=1,
                    help='batches of 16 windows per forward call (grouped launch: one launch per layer for all of them, own batch-norm statistics per batch; rows unchanged)')
    ap.add_argument('--partial_batch', choices=['drop', 'pad'], default='drop',
                    help="trailing windows that do not fill a batch of 16: 'drop' (the reference's queue never dequeues them) or 'pad' with zero windows")
    ap.add_argument('--power_maps', action='store_true',
                    help='also compute the per-sample directional RMS maps of prediction and ground truth on the device (the inputs of the '
                         "reference's EMD metric, eval.py:188-191) and save this rank's maps to <model_dir>/eval-powermaps-rank<r>.npz")
    args = ap.parse_args(argv)
    means, count = evaluate(args.model_dir, args.db_dir, args.subset_fn, args.layouts_fn, overwrite=args.overwrite,
                            partial_batch=args.partial_batch, power_maps=args.power_maps, groups=args.groups)
    if args.power_maps and evaluate.last_maps:
        maps = evaluate.last_maps                      # [pred, gt, pred, gt, ...] per batch, each [n, 7, 12]
        np.savez(os.path.join(args.model_dir, 'eval-powermaps-rank%d.npz' % int(os.environ.get('RANK', 0))),
                 pred=np.concatenate(maps[0::2], 0), gt=np.concatenate(maps[1::2], 0))
    if int(os.environ.get('RANK', 0)) == 0:
        print('EVAL | %d samples' % count)
        for k, v in means.items():
            print('EVAL | \t %s %f' % (k, v))


if __name__ == '__main__':
    main()
