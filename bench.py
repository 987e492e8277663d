#!/usr/bin/env python
"""Headline benchmark: ambisonic seconds generated per second (0.1 s windows, 224x448 video).

    python bench.py --gpus N --steps K --warmup W [--config av|a|avf|eval]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
            --master-port P bench.py --gpus N --steps K --warmup W)

One "step" = one pass of the hot path (sagen_forward: STFT -> audio [+ ResNet18 video / flow] encoders -> U-Net mask
decoder -> iSTFT -> ambisonic mix) over one batch of synthetic 0.1 s windows per GPU.  Inputs are resident in HBM before the
timed region.  Windows / batches shard over ranks with no data-path collective (the only collective is the eval-style metric
all-reduce, RCCL, once at the end of the timed region).  Rank 0 prints ONE JSON line.

--config selects the BASELINE.json configuration:
    av    configs[1]  audio + video encoders, 32 windows per batch                      (default: the configuration the metric is quoted on)
    a     configs[0]  audio encoder only, deploy.py's batch of 10 windows               (the reference's CPU-runnable plumbing case, here on the GPU)
    avf   configs[2]  audio + video + flow encoders, 32 windows per batch
    train configs[4]  train.py loop: audio + video encoders, batch 32 per GPU, one optimiser step per "step" = forward with retained
                      activations + stft loss + full backward + gradient all-reduce over the ranks (RCCL, flat buckets) + fused Adam +
                      filter re-pack; value = ambisonic seconds TRAINED per second (weak scaling)
    eval  configs[3]  YT-All stand-in: 1024 synthetic clips x 9 windows, batches of 16 cut from one global window order and dealt
                      whole to the ranks (spatialaudiogen_amd.evaluate.batch_shard), forward + on-device evaluation_ops per batch,
                      ONE all-reduce of the metric sums at the end; strong scaling (total work fixed); --steps caps the batches per rank
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# needed-only algorithmic work per window (BASELINE.md 2 / SURVEY.md 8d) and the workload of each configuration
CONFIGS = {
    'av': dict(encoders=['audio', 'video'], batch=32, gflop=8.41, scaling='weak',
               workload='configs[1]: audio+video encoders (no flow), 224x448@10fps + 48 kHz mono, batch 32 x 0.1 s windows per GPU, '
                        'FREQ_MASK separation, 32 tracks'),
    'a': dict(encoders=['audio'], batch=10, gflop=1.13, scaling='weak',
              workload="configs[0]: audio-only encoder, deploy.py's batch of 10 x 0.1 s windows per GPU (the reference's CPU plumbing "
                       'case run on the GPU; synthetic weights, the REC-Street checkpoint is not available offline)'),
    'avf': dict(encoders=['audio', 'video', 'flow'], batch=32, gflop=15.69, scaling='weak',
                workload='configs[2]: audio+video+flow encoders (three-stream fusion), batch 32 x 0.1 s windows per GPU'),
    'eval': dict(encoders=['audio', 'video'], batch=16, gflop=8.41, scaling='strong',
                 workload='configs[3]: YT-All eval stand-in, 1024 synthetic clips x 9 windows (every 10th window of a 10 s clip), batches of '
                          '16 dealt whole to the ranks, forward + on-device evaluation metrics, one metric all-reduce'),
}
CONFIGS['train'] = dict(encoders=['audio', 'video'], batch=32, gflop=8.41, scaling='weak',
                        workload='configs[4]: train.py loop, audio+video encoders, batch 32 x 0.1 s windows per GPU, Adam step (the reference '
                                 'optimiser, myutils.py:220) with gradient all-reduce across the GPUs, synthetic batches')
EVAL_CLIPS, EVAL_WINDOWS_PER_CLIP = 1024, 9
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0         # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16: 32 cycles / SIMD)
# the bf16x3 kernels evaluate every fp32 product as 6 bf16 products (3-way operand split, fp32 accumulate): their matrix roof
# in ALGORITHMIC fp32 FLOP/s is the bf16 peak / 6
PEAK_BF16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
# conv3h_kernel (the trunk's 3x3 convs since round 4): two fp16 planes per operand, 3 products per fp32 multiply on v_mfma_f32_32x32x16_f16
# (same rate as bf16): roof = 2500 / 3 in algorithmic fp32 FLOP/s
PEAK_FP16X2_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0
# Context, not the roof: a kernel issuing nothing but v_mfma_f32_32x32x16_bf16 on every SIMD sustains 1.66-1.81 PFLOP/s with normally
# distributed operands (power management; 2.28-2.48 with all-zero operands) - tools/probe/mfma_peak.py, profiles/r03_mfma_sustained.txt
SUSTAINED_BF16X3_TFLOPS = 1740.0 / 6.0
SUSTAINED_NOTE = ('achieved / 290 TFLOP/s = the fp32-equivalent rate a pure v_mfma_f32_32x32x16_bf16 loop sustains chip-wide on '
                  'normally distributed operands (1.66-1.81 PFLOP/s bf16: profiles/r03_mfma_sustained.txt); `frac` above is against the data-sheet peak')


DEFAULT_GROUP = 0          # 0 = auto: see pick_group (8 .. 16 batches per call in an even number of calls, else the largest divisor of K up to 10)
DEFAULT_IN_FLIGHT = 0      # 0 = auto: two contexts in flight with grouped launches, three with one batch per call (measured below)
MAX_AUTO_GROUP = 10
MAX_PAIRED_GROUP = 16      # (larger groups measured no further gain; the native limit of sagen_create_grouped is 32)


def pick_group(steps, asked, cap=MAX_AUTO_GROUP):
    """Batches per grouped launch.  Measured on one box (tools/ab_group2.sh, configs[1], ambisonic-s/s, first region / median of the repeats;
    K = 20 | 30 steps): 1 x 3 contexts 2 631 / 2 685 | 2 667 / 2 700; 2 x 2: 2 720 / 2 790 | 2 786 / 2 823; 4 x 2: 2 839 / 2 908 | 2 861 / 2 950;
    5 x 2: 2 848 / 2 960 | 2 919 / 2 990; 5 x 3: 2 808 / 2 940 | 2 936 / 3 030; 6 x 2: - | 2 959 / 3 025; 10 x 2: 2 986 / 3 075 | 2 967 / 3 015;
    10 x 1: 2 881 / 2 920 | 2 907 / 2 935 - the fixed cost of a launch is paid once per group, and two grouped contexts still fill each
    other's tails.  Auto: the largest G in 8 .. 16 that splits K into an EVEN number of calls (below); else the largest divisor of K up to 10
    (no remainder call in the timed region); a K without a divisor of at least 4 takes min(K, 10) and runs the remainder as one smaller call."""
    if asked > 0:
        return asked
    if steps <= 0:
        return 1
    # two contexts in flight want an EVEN number of calls (an unpaired last call runs without a partner), and larger groups keep paying
    # (K = 30, same box: 10 per call x 3 calls 3 133 - 3 150, 15 x 2 calls 3 279 - 3 286; K = 32: 16 x 2 3 251 - 3 272): up to 16 per call then
    paired = [g for g in range(8, MAX_PAIRED_GROUP + 1) if steps % g == 0 and (steps // g) % 2 == 0]
    if paired:
        return max(paired)
    best = max(g for g in range(1, cap + 1) if steps % g == 0)
    return best if best >= 4 or best == steps else max(1, min(cap, steps))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='av')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-legs', action='store_true', help='skip the one-in-flight and H2D-inclusive legs (they run after the timed region)')
    ap.add_argument('--in-flight', type=int, default=DEFAULT_IN_FLIGHT,
                    help='batches in flight per GPU: steps rotate over this many native contexts, each on its own stream, the streams '
                         'probed to really run concurrently (DESIGN.md 6.1); 1 = strictly one forward at a time (round 4: three, measured '
                         '2 210 against 2 083-2 131 ambisonic-s/s with two on one box - the fp16x2 kernels leave LDS and registers for a third '
                         "batch's workgroups; four = three)")
    ap.add_argument('--no-other-configs', action='store_true',
                    help='default run (config av, one GPU): skip the short legs of the other BASELINE configurations (a, avf, eval, train) that '
                         'follow the headline, each a fresh process of this script whose line is attached as `leg_<config>`')
    ap.add_argument('--no-autotune', action='store_true', help='use shape heuristics instead of the timed per-layer plan')
    ap.add_argument('--plan-file', default=None, help='replay this saved launch plan if it exists, else autotune and save it')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='budget of the CPU baseline leg')
    ap.add_argument('--group', type=int, default=DEFAULT_GROUP,
                    help='batches per grouped launch (include/sagen.h: sagen_forward_grouped): each native context carries this many INDEPENDENT batches '
                         'of the configuration per forward call, one launch per layer - every batch keeps its own batch-norm statistics and its '
                         'output is bit-identical to a forward of that batch alone (tests/test_gpu_grouped.py); a step is still ONE batch: K steps = '
                         'K / group calls (+ one smaller call for a remainder).  1 = one batch per call (rounds 1-5)')
    ap.add_argument('--no-repeats', action='store_true', help='skip the four extra timed regions behind the headline (headline_repeats)')
    ap.add_argument('--no-pmc', action='store_true',
                    help='do not measure roofline.traffic in this run (two short child processes of this script under rocprofv3 --pmc, when '
                         'rocprofv3 is on the box); the committed profiles/pmc_traffic.json is quoted instead')
    ap.add_argument('--pmc-child', action='store_true', help=argparse.SUPPRESS)      # internal: a few forwards of the saved plan, no report
    ap.add_argument('--float-frames', action='store_true',
                    help='keep the resident video frames as the float32 the feeder would make of them (x/255 - 0.5) instead of the uint8 the '
                         'JPEG decoder produced: the general float entry point (sagen_forward) in the timed region')
    return ap.parse_args()


def cpu_baseline(P, inputs, budget_s, encoders, batch):
    """The reference-equivalent CPU path (oracle/torch_ref.py, torch-CPU/oneDNN fp32, all host
    threads) on the same synthetic batch; TF 1.4 cannot be installed offline."""
    import torch
    from oracle.torch_ref import TorchRef
    threads = torch.get_num_threads()
    ref = TorchRef(P, encoders, dtype=torch.float32)
    args = [inputs['audio']] + [inputs[k] for k in ('video', 'flow') if k in inputs]
    t0 = time.time()
    ref.forward(*args)                                 # warm-up (oneDNN primitive creation)
    warm = time.time() - t0
    times = []
    t_start = time.time()
    while len(times) < 2 or (time.time() - t_start < budget_s and len(times) < 20):
        t0 = time.time()
        ref.forward(*args)
        times.append(time.time() - t0)
    med = float(np.median(times))
    return {
        'value': round(0.1 * batch / med, 3), 'unit': 'ambisonic-s/s', 'cores': threads, 'kind': 'port',
        'sample': '%d timed batches of %d windows (same synthetic %s workload, fp32), median %.3f s/batch, '
                  'warm-up %.2f s; torch-CPU/oneDNN stand-in for the TF1 CPU path (TF1 unavailable offline)'
                  % (len(times), batch, '+'.join(encoders), med, warm),
    }


def cpu_baseline_train(P, inputs, target, budget_s, encoders, batch):
    """One training iteration of the reference-equivalent CPU path: forward + stft loss + autograd backward of
    oracle/torch_ref.py in fp32 on all host threads (the optimiser update is not included - it is noise next to the backward)."""
    import torch
    from oracle.torch_ref import TorchRef
    threads = torch.get_num_threads()
    ref = TorchRef(P, encoders, dtype=torch.float32)
    a = (inputs['audio'], inputs.get('video'), inputs.get('flow'), target)
    t0 = time.time()
    ref.loss_and_grads(*a)
    warm = time.time() - t0
    times = []
    t_start = time.time()
    while len(times) < 1 or (time.time() - t_start < budget_s and len(times) < 10):
        t0 = time.time()
        ref.loss_and_grads(*a)
        times.append(time.time() - t0)
    med = float(np.median(times))
    return {'value': round(0.1 * batch / med, 3), 'unit': 'ambisonic-s/s', 'cores': threads, 'kind': 'port',
            'sample': '%d timed training iterations (forward + loss + autograd backward, no optimiser) of %d windows, same synthetic %s '
                      'batch, fp32, median %.2f s/iteration, warm-up %.2f s; torch-CPU/oneDNN stand-in for the TF1 CPU path'
                      % (len(times), batch, '+'.join(encoders), med, warm)}


def other_config_legs(args):
    """BASELINE configs[0], [2], [3], [4] next to the headline (configs[1]) in the default one-GPU run: each leg is `bench.py --config X`
    in a fresh process AFTER the headline's timed region and extra legs (nothing of this process is running on the GPU any more), with
    its own warm-up / autotune / timed region under the same contract; its line is reduced to value, ms_per_step, steps and roofline.
    A leg that fails is recorded as such - the headline never depends on it."""
    import subprocess
    legs = {}
    # (timed regions of 45 - 90 ms: the 9 ms that 60 steps of the audio-only configuration last moved by +-20 % from run to run)
    plan = {'a': (300, 30), 'avf': (40, 6), 'eval': (100, 10), 'train': (14, 3)}
    for name, (steps, warm) in plan.items():
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--config', name, '--steps', str(steps), '--warmup', str(warm),
               '--no-cpu-baseline', '--no-extra-legs', '--no-other-configs'] + (['--no-autotune'] if args.no_autotune else [])
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            if r.returncode != 0 or not line:
                legs['leg_' + name] = {'error': 'exit code %d: %s' % (r.returncode, (r.stderr or r.stdout)[-400:])}
                continue
            doc = json.loads(line[-1])
            rf = doc.get('roofline', {})
            legs['leg_' + name] = {
                'metric': doc['metric'], 'value': doc['value'], 'unit': doc['unit'], 'ms_per_step': doc['ms_per_step'], 'steps': doc['steps'],
                'warmup': doc['warmup'], 'scaling': doc['scaling'], 'workload': doc['config']['workload'],
                'batches_in_flight': doc['config'].get('batches_in_flight'), 'windows_per_gpu_per_step': doc['config'].get('windows_per_gpu_per_step'),
                'roofline': {k: rf.get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_us', 'launches_per_step',
                                                    'share_of_step_time', 'whole_step')},
                'wall_s': round(time.time() - t0, 1)}
        except Exception as e:                                    # noqa: BLE001 (a leg must never take the headline down)
            legs['leg_' + name] = {'error': repr(e)[:400]}
    return legs


def init_ranks(dist, world, rank):
    """One process per GPU over RCCL: `nccl` IS RCCL on ROCm.  SAGEN_DIST_BACKEND=gloo is the test hook for several ranks on ONE GPU
    (tests/test_gpu_bench.py); anything else than what was asked for is an error, not a silent fallback."""
    backend = os.environ.get('SAGEN_DIST_BACKEND', 'nccl')
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group(backend, rank=rank, world_size=world)
        got = dist.get_backend()
        assert got == backend, 'process group runs on %r, asked for %r' % (got, backend)
        if 'SAGEN_DIST_BACKEND' not in os.environ:
            assert got == 'nccl', 'the multi-GPU path must run over RCCL (backend nccl), got %r' % got
    return backend


def ranks_report(dist, torch, world, backend, elapsed_local, steps):
    """What an 8-GPU run must show without anyone looking at the node: every rank took part (all-reduce of ones), which backend
    carried it, and the spread of the per-rank step times (the headline uses the max)."""
    if world <= 1:
        return {'backend': None, 'ranks_seen': 1, 'rank_ms_per_step': {'min': round(1e3 * elapsed_local / max(steps, 1), 4), 'max': round(1e3 * elapsed_local / max(steps, 1), 4)}}
    dev = 'cuda' if backend != 'gloo' else 'cpu'
    ones = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(ones)
    t = torch.zeros(world, dtype=torch.float64, device=dev)
    t[dist.get_rank()] = 1e3 * elapsed_local / max(steps, 1)
    dist.all_reduce(t)
    per = [round(float(x), 4) for x in t.cpu()]
    return {'backend': dist.get_backend() + (' (RCCL)' if dist.get_backend() == 'nccl' else ' (test hook: several ranks on one GPU)'),
            'ranks_seen': int(round(float(ones.item()))), 'rank_ms_per_step': {'min': min(per), 'max': max(per), 'per_rank': per},
            'gpus_visible_to_rank0': torch.cuda.device_count()}


def main_train(args, cfg):
    """--config train: BASELINE configs[4]."""
    ENCODERS, BATCH = cfg['encoders'], cfg['batch']
    import torch
    import torch.distributed as dist
    from spatialaudiogen_amd.model import SptAudioGen
    from spatialaudiogen_amd.train import Trainer
    from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the hot path has no CPU implementation')
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    backend = init_ranks(dist, world, rank)
    P = init_weights(variable_specs(ENCODERS), seed=0, mode='bench')           # same replica on every rank
    inp = synth_inputs(BATCH, ENCODERS, seed=1234 + rank)                      # every rank trains on its own windows
    target = (inp['audio'][:, 24000:28800, :] * np.array([0.5, 0.25, -0.5], np.float32)).astype(np.float32)
    net = SptAudioGen(1, encoders=ENCODERS, separation='unet_mask')
    net.load_variables(P)
    tr = Trainer(net, batch=BATCH, lr=1e-4, lr_iters=250000, lr_decay=0.5)     # train.py defaults
    dev = [torch.as_tensor(inp[k]).cuda() if k in inp else None for k in ('audio', 'video', 'flow')] + [torch.as_tensor(target).cuda()]
    if dev[1] is not None and not args.float_frames:
        # frames as the training feeder decodes them (uint8; x / 255 - 0.5 on the device, sagen_train_step_u8): the default entry point
        dev[1] = torch.round((dev[1].double() + 0.5) * 255.0).clamp(0, 255).to(torch.uint8)

    # launch plan: tuned on rank 0, broadcast, so that every rank runs the same kernels
    plan = []
    if not args.no_autotune:
        if args.plan_file and os.path.exists(args.plan_file):
            plan = json.load(open(args.plan_file))['plan']
        elif rank == 0:
            plan = tr.autotune(*dev)
            if args.plan_file:
                json.dump({'batch': BATCH, 'encoders': ENCODERS, 'plan': plan}, open(args.plan_file, 'w'), indent=1)
        if world > 1:
            box = [plan]
            dist.broadcast_object_list(box, src=0)
            plan = box[0]
        tr.load_plan_rows(plan)
        got = sorted((l, t, k) for l, t, k, _ in tr.plan())
        want = sorted((l, t if not l.endswith('#materialize') else '-', k) for l, t, k, _ in plan)
        assert got == want, 'rank %d runs a different launch plan than rank 0' % rank
        # the tuning step left meaningless gradients / optimiser state untouched (no apply): nothing to undo
    torch.cuda.synchronize()

    def step():
        return tr.step(*dev)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, lr = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    last_loss = float(loss)
    ranks = ranks_report(dist, torch, world, backend, elapsed, args.steps)
    assert ranks['ranks_seen'] == world, ranks
    comm = {}
    if world > 1 and tr.bucket_events is not None:      # one more step with timing events on the communication stream
        tr.step(*dev, comm_timing=comm)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if backend != 'gloo' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    steps = args.steps
    value = 0.1 * BATCH * world * steps / elapsed
    ms_per_step = 1e3 * elapsed / max(steps, 1)

    # per-launch HIP events of the native runtime over two more steps (forward + backward launches; the optimiser and the
    # all-reduce are torch-side and appear only in the step time)
    tr.profile_enable(True)
    agg, by_phase = {}, {}
    nprof = 2
    for _ in range(nprof):
        tr.forward_backward(*dev)
        for k, layer, us, fl in tr.profile_report():
            a = agg.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1; a[1] += us; a[2] += fl
            ph = 'wgrad' if layer.startswith('wgrad:') else 'dgrad' if layer.startswith('dgrad:') else \
                'bwd-elementwise' if (':' in layer.split('/')[0] or layer.endswith(':bwd')) else 'forward'
            b = by_phase.setdefault(ph, [0.0, 0.0])
            b[0] += us; b[1] += fl
    tr.profile_enable(False)
    total_us = sum(a[1] for a in agg.values())
    total_fl = sum(a[2] for a in agg.values())
    dom = max(agg, key=lambda k: agg[k][1])
    n_l, us_l, fl_l = agg[dom]
    achieved = fl_l / (us_l * 1e-6) / 1e12
    h2 = dom.startswith('conv3h') or dom.startswith('wgrad3h') or (dom.startswith('conv3g') and dom.rstrip('>').endswith('true'))
    b3 = not h2 and (dom.startswith('igemm3') or dom.startswith('conv3p') or dom.startswith('wgrad3') or dom.startswith('conv3g'))
    peak = PEAK_FP16X2_TFLOPS if h2 else (PEAK_BF16X3_TFLOPS if b3 else PEAK_FP32_MFMA_TFLOPS)
    traffic, traffic_src = None, None
    try:     # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/)
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic_train.json')) as f:
            traffic = json.load(f).get(dom)
        if traffic is not None:
            traffic_src = ('profiles/pmc_traffic_train.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (FETCH_SIZE x2 per the '
                           'gfx950 note), launch-weighted mean, NOT measured in this run' % committed_rev('profiles/pmc_traffic_train.json'))
    except Exception:
        pass
    step_tflops = total_fl / nprof / (ms_per_step * 1e-3) / 1e12
    roofline = {
        'bound': 'mfma', 'kernel': dom, 'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
        'frac': round(achieved / peak, 4), 'traffic': traffic, 'traffic_source': traffic_src,
        'frac_of_sustained_mfma_rate': round(achieved / SUSTAINED_BF16X3_TFLOPS, 4) if b3 else None, 'sustained_note': SUSTAINED_NOTE if b3 else None,
        'peak_basis': ('dense fp16 MFMA peak 2500 TF / 3 products per fp32 multiply (fp16x2 planes); algorithmic fp32 FLOPs' if h2 else
                       'dense bf16 MFMA peak 2500 TF / 6 products per fp32 multiply (bf16x3 kernel); algorithmic fp32 FLOPs'
                       if b3 else 'fp32 MFMA peak (v_mfma_f32_32x32x2_f32)'),
        'fp16x2_families_us_per_step': round(sum(a[1] for k, a in agg.items() if k.startswith(('conv3h', 'wgrad3h')) or (k.startswith('conv3g') and k.rstrip('>').endswith('true'))) / nprof, 1),
        'bf16x3_families_us_per_step': round(sum(a[1] for k, a in agg.items() if k.startswith(('igemm3', 'conv3p', 'wgrad3_', 'wgrad3r')) or (k.startswith('conv3g') and not k.rstrip('>').endswith('true'))) / nprof, 1),
        'achieved_over_fp32_mfma_peak': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
        'launches_per_step': n_l // nprof, 'avg_launch_us': round(us_l / n_l, 2), 'gflop_per_launch': round(fl_l / n_l / 1e9, 3),
        'share_of_step_time': round(us_l / total_us, 3),
        'phases_us_per_step': {k: round(v[0] / nprof, 1) for k, v in sorted(by_phase.items())},
        'phases_tflops': {k: round(v[1] / (v[0] * 1e-6) / 1e12, 1) for k, v in sorted(by_phase.items()) if v[1] > 0},
        'whole_step': {'achieved': round(step_tflops, 2), 'frac': round(step_tflops / PEAK_BF16X3_TFLOPS, 4),
                       'frac_basis': 'all launched fp32-equivalent FLOPs of the step / step time, against the six-product bf16x3 roof (417 TFLOP/s); the '
                                     "stride-1 3x3 trunk convs (forward, data and weight gradients: 39 of the step's contractions) run on "
                                     'the three-product fp16x2 kernels, the rest on bf16x3',
                       'achieved_over_fp32_mfma_peak': round(step_tflops / PEAK_FP32_MFMA_TFLOPS, 4),
                       'gflop_per_window_launched': round(total_fl / nprof / BATCH / 1e9, 2),
                       'gflop_per_window_forward_needed_only': cfg['gflop'],
                       'kernel_time_us_per_step': round(total_us / nprof, 1)},
    }
    result = {
        'metric': 'ambisonic seconds trained/sec (0.1 s windows, 224x448 video; one Adam step per batch)',
        'value': round(value, 2), 'unit': 'ambisonic-s/s', 'n_gpus': world, 'steps': steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 (fp32 in / out / accumulate and fp32 Adam state; products on the 16-bit matrix cores with fp32-equivalent operand splits: '
                 'two fp16 planes (3 products per multiply) for the forward, data gradient and weight gradient of the stride-1 3x3 trunk convs, '
                 'three bf16 planes (6 products) elsewhere - every gradient against fp64 autograd in tests/test_gpu_backward.py)',
        'data': 'synthetic',
        'config': {'workload': cfg['workload'], 'name': 'train', 'windows_per_gpu_per_step': BATCH,
                   'video_frames': 'float32' if args.float_frames else 'uint8 as decoded (x / 255 - 0.5 applied on the device), sagen_train_step_u8',
                   'windows_per_s': round(BATCH * world * steps / elapsed, 1), 'optimizer': 'Adam (lr 1e-4), fused over %d flat buckets' % len(tr.opt.params),
                   'gradient_exchange': 'sum all-reduce of %d buckets (%.0f MB) per step over %d rank(s), %s' % (
                       len(tr.opt.grads), sum(g.numel() for g in tr.opt.grads) * 4 / 1e6, world, backend if world > 1 else 'none'),
                   'launch_plan': 'autotuned on rank 0 and broadcast (%d contractions)' % len(plan) if plan else 'shape heuristics',
                   'weights': 'random init (Xavier / BN identity), same replica on every rank', 'last_loss': last_loss},
        'ranks': ranks,
        'gradient_exchange_timing': comm or None,
        'roofline': roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline_train(P, inp, target, args.cpu_seconds, ENCODERS, BATCH)
        cb['gpu_over_cpu'] = round(value / cb['value'], 1)
        result['cpu_baseline'] = cb
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def committed_rev(path):
    """'commit <short hash> <date>' of the last commit that touched `path` (so a quoted file says how old it is), or 'uncommitted'."""
    import subprocess
    try:
        out = subprocess.run(['git', '-C', ROOT, 'log', '-1', '--format=%h %cs', '--', path], capture_output=True, text=True, timeout=10).stdout.strip()
        return 'commit ' + out if out else 'not under version control here'
    except Exception:
        return 'revision unknown'


def short_kernel_name(name):
    name = name.replace('void sagen::', '').replace('sagen::', '')
    return name.split('(')[0].replace(' ', '')


def pmc_traffic_in_run(args, net, batch, dom):
    """roofline.traffic measured IN THIS RUN: HBM bytes per launch of the dominant kernel from the memory-side counters, collected as
    MI355X_MICROARCH.md prescribes - FETCH_SIZE and WRITE_SIZE in SEPARATE `rocprofv3 --pmc` passes (nothing else enabled: no
    kernel / sys / hip trace in the same run), FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads), both x 1024.
    Each pass is a child process of this script (`--pmc-child`: three forwards of the plan this run timed, one context).  Returns
    (bytes, source) or (None, None) when rocprofv3 is not on the box / a pass fails - the caller then quotes the committed file."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if prof is None or os.environ.get('BENCH_NO_PMC'):
        return None, None
    tmp = tempfile.mkdtemp(prefix='sagen_pmc_', dir='/tmp')
    try:
        plan_fn = os.path.join(tmp, 'plan.json')
        net.save_plan(batch, plan_fn)
        vals = {}
        env = dict(os.environ, TMPDIR='/tmp', BENCH_NO_PMC='1')
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, ctr)
            cmd = [prof, '--pmc', ctr, '--output-format', 'csv', '-d', out, '--', sys.executable, os.path.abspath(__file__), '--pmc-child',
                   '--config', args.config, '--in-flight', '1', '--group', str(pick_group(args.steps, args.group)), '--steps', str(args.steps), '--plan-file', plan_fn, '--no-cpu-baseline', '--no-other-configs'] + \
                  (['--float-frames'] if args.float_frames else [])
            r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=240)
            fs = glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not fs:
                return None, None
            v = [float(row['Counter_Value']) for f in fs for row in csv.DictReader(open(f))
                 if row['Counter_Name'] == ctr and short_kernel_name(row['Kernel_Name']).startswith(dom.rstrip('>'))]
            if not v:
                return None, None
            vals[ctr] = (sum(v) / len(v), len(v))
        byts = (2.0 * vals['FETCH_SIZE'][0] + vals['WRITE_SIZE'][0]) * 1024.0
        return int(round(byts)), ('measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE as two separate child passes of this command '
                                  '(three forwards each, one context, the plan timed above; mean over %d / %d launches of %s); bytes = (2 x FETCH_SIZE + '
                                  'WRITE_SIZE) x 1024 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports half of wide coalesced reads)'
                                  % (vals['FETCH_SIZE'][1], vals['WRITE_SIZE'][1], dom))
    except Exception:
        return None, None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    args = parse()
    cfg = CONFIGS[args.config]
    ENCODERS, BATCH = cfg['encoders'], cfg['batch']
    if os.environ.get('BENCH_PROBE_BATCH'):
        # experiment switch (NOT the metric's configuration; the line says so in config.workload): another batch size per step, e.g.
        # 96 = what three batches of 32 grouped into one launch per layer would cost (grouped launch)
        BATCH = int(os.environ['BENCH_PROBE_BATCH'])
        cfg = dict(cfg, batch=BATCH, workload='PROBE (batch %d instead of %d, not the configuration the metric is quoted on): ' % (BATCH, cfg['batch']) + cfg['workload'])
    if args.gpus > 1 and 'RANK' not in os.environ:
        # convenience: re-launch ourselves under torchrun, one rank per GPU
        import subprocess
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', os.environ.get('MASTER_PORT', '29533'),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if args.config == 'train':
        return main_train(args, cfg)

    if args.in_flight != 1:
        # several batches in flight: each context stays on its caller's stream (the other batch is the overlap)
        if not os.environ.get('BENCH_TWO_STREAM_CONTEXTS'):      # (experiment switch: keep each context's own second stream as well)
            os.environ['SAGEN_ONE_STREAM'] = '1'
    import torch
    import torch.distributed as dist
    from spatialaudiogen_amd.model import SptAudioGen
    from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the hot path has no CPU implementation')
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    backend = init_ranks(dist, world, rank)

    P = init_weights(variable_specs(ENCODERS), seed=0, mode='bench')           # same replica on every rank
    is_eval = args.config == 'eval'
    # eval: a pool of distinct synthetic windows stands in for the 9216 windows of the clip set (window w of the global order
    # uses pool entry w % pool); the other configurations: each rank owns one batch of its own windows
    # (the other configurations cycle over three distinct batches of this rank's own windows: no step re-reads the inputs of the one before)
    # which batches does this rank run?  eval: its whole-batch shard of the global order; otherwise its own batches, repeated
    if is_eval:
        from spatialaudiogen_amd.evaluate import batch_shard
        lo_b, hi_b, n_batches = batch_shard(EVAL_CLIPS * EVAL_WINDOWS_PER_CLIP, rank, world, 'drop')
        eval_batches = list(range(lo_b, hi_b))
        eval_steps = min(args.steps, len(eval_batches)) if args.steps > 0 else len(eval_batches)
        # eval: consecutive batches of the shard per call, a divisor of the step count only (no remainder call: the metrics follow the batches)
        G = args.group if args.group > 0 and eval_steps % args.group == 0 else max(g for g in range(1, MAX_AUTO_GROUP + 1) if eval_steps % g == 0)
    else:
        G = pick_group(args.steps, args.group)          # batches per grouped call
    CALLB = G * BATCH                                   # windows per forward call
    NPOOL = max(4, G) if is_eval else (3 if G == 1 else 2 * G)  # resident batches (G > 1: two distinct input sets per call, alternated)
    POOL = NPOOL * BATCH
    inp = synth_inputs(POOL, ENCODERS, seed=1234 + (0 if is_eval else rank))
    # Steps are independent batches, so NF of them are kept in flight: step i runs on native context i % NF and stream
    # i % NF (each context has its own workspace and its own second stream).  Every step is still one full forward of
    # one batch; the kernels of neighbouring steps fill each other's launch tails.  Outputs are bit-identical to
    # strictly sequential execution (tests/test_gpu_streams.py, tools/two_in_flight.py).
    NF = args.in_flight if args.in_flight > 0 else (2 if G > 1 else 3)
    nets = [SptAudioGen(1, encoders=ENCODERS, separation='unet_mask', groups=G) for _ in range(NF)]
    for n in nets:
        n.load_variables(P)
    net = nets[0]
    if G > 1 and BATCH < 16 and not os.environ.get('BENCH_NO_SMALL_BATCH_DECODER_PLANES'):
        # the scatter-form decoder contracts fp16x2 planes from batch 16 on (a single batch of 10 does not pay for the four pack launches:
        # DESIGN.md 3.2); inside a grouped call the packs are amortised over the groups, so the small-batch configuration takes the planes too
        for n in nets:
            n.set_option(BATCH, 'decoder_planes', 1)
    streams = []                        # created after the contexts (below): ROCm maps HIP streams to hardware queues in creation order
    dev_in = {k: torch.as_tensor(v).cuda() for k, v in inp.items()}
    if is_eval:     # window w of the global order uses pool entry w % POOL: the pool twice in a row makes any run of <= POOL windows one slice
        dev_in = {k: torch.cat([v, v], 0) for k, v in dev_in.items()}
    names_in = ['audio'] + [k for k in ('video', 'flow') if k in dev_in]
    # Video frames are resident the way the JPEG decoder produced them: uint8 (feeder.py reads .jpg; myutils.py:88-89 normalises with
    # x / 255 - 0.5).  sagen_forward_u8 applies that normalisation on the device; the synthetic frames are exact images of uint8 values
    # (checked here), so both entry points see the same pixels.  --float-frames keeps float32 frames (the general float entry point).
    frames_note = 'float32 (x / 255 - 0.5 applied by the host), sagen_forward'
    float_video = dev_in.get('video')
    if 'video' in dev_in and not args.float_frames:
        t = dev_in['video']
        u8 = torch.round((t.double() + 0.5) * 255.0).clamp(0, 255).to(torch.uint8)
        assert bool(((u8.double() / 255.0 - 0.5).float() == t).all()), 'synthetic frames are not exact images of uint8 values'
        dev_in['video'] = u8
        frames_note = 'uint8 as decoded (x / 255 - 0.5 applied on the device), sagen_forward_u8'

    def batch_inputs(b):                # device views of the windows of call b's G batches (eval: of the G batches from global batch b on)
        lo = (b * (BATCH if is_eval else CALLB)) % POOL
        return [dev_in[k][lo:lo + CALLB] for k in names_in]

    outs = [torch.empty(CALLB, 4800, 3, device='cuda') for _ in range(NF)]
    metric = torch.zeros(14, dtype=torch.float64, device='cuda')
    eval_sums = [torch.zeros(12, dtype=torch.float64, device='cuda') for _ in range(NF)]
    counter = [0]

    if is_eval:
        my_batches, steps = eval_batches[::G], eval_steps             # (the first batch of every call)
    else:
        my_batches, steps = list(range(NPOOL // G)), args.steps
    n_calls, rem = steps // G, steps % G                # K steps (batches) = n_calls grouped calls + one call of the `rem` left over

    # (built ONCE: a torch.tensor(list, device='cuda') inside the step is a blocking pageable-host copy behind the forward just
    #  enqueued on the same stream - it made the launching thread wait for every forward and took the batches out of flight)
    tgt_scale = torch.tensor([0.5, 0.25, -0.5], device='cuda') if is_eval else None

    def run_step(j, b, ctx_nets, ctx_outs):
        """one forward (+ metrics in eval mode) of global batch b on context j (current stream)"""
        a = batch_inputs(b)
        ctx_nets[j].inference_ops(*a, out=ctx_outs[j])
        if is_eval:         # targets: a fixed pseudo ground truth (the W-channel crop scaled per channel) - the metric kernels run at full cost
            tgt = a[0][:, 24000:28800, :] * tgt_scale
            ps, _ = ctx_nets[j].evaluation_ps(ctx_outs[j], tgt.contiguous())
            eval_sums[j] += ps.double().sum(1).reshape(-1)

    def step():
        i = counter[0]
        j = i % NF
        counter[0] += 1
        b = my_batches[i % len(my_batches)]
        if NF == 1:
            run_step(0, b, nets, outs)
        else:
            with torch.cuda.stream(streams[j]):
                run_step(j, b, nets, outs)
        return j

    def barrier():
        if world > 1:
            dist.barrier()

    def reduce_metric():
        # eval-style metric reduction: per-rank sums + count, one all-reduce (RCCL)
        if NF > 1:                          # the reduction (current stream) consumes what the step streams produced
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
        metric[0] = sum((o.double() ** 2).sum() for o in outs) / NF
        metric[1] = float(BATCH)
        metric[2:] = sum(eval_sums)
        if world > 1:
            if backend == 'gloo':
                m = metric.cpu()
                dist.all_reduce(m)
                metric.copy_(m)
            else:
                dist.all_reduce(metric)

    # untimed set-up: per-layer (tile, split-K) autotune on this rank's own batch, then W warm-up steps
    plan = []
    a0 = batch_inputs(my_batches[0] if my_batches else 0)
    if args.plan_file and os.path.exists(args.plan_file):
        net.inference_ops(*a0, out=outs[0])
        net.load_plan(BATCH, args.plan_file)
        plan = net.plan(BATCH)
    elif not args.no_autotune:
        # tuned on rank 0 and broadcast: every rank (and every context in flight) runs the SAME kernels - independent tunings can
        # settle on different tiles for near-ties, which would make the ranks' step times (and rounding) differ
        if rank == 0:
            plan = net.autotune(*a0)
            if args.plan_file:
                net.save_plan(BATCH, args.plan_file)
        if world > 1:
            box = [plan]
            dist.broadcast_object_list(box, src=0)
            plan = [tuple(r) for r in box[0]]
            if rank != 0:
                net.inference_ops(*a0, out=outs[0])           # creates the context
                tn = SptAudioGen.tile_names()
                for layer, tile, sk, _ in plan:
                    net.plan_set(BATCH, layer, tn.index(tile) if tile in tn else 0, sk)
            mine = sorted((l, t, k) for l, t, k, _ in net.plan(BATCH))
            assert mine == sorted((l, t, k) for l, t, k, _ in plan), 'rank %d runs a different launch plan than rank 0' % rank
    names = SptAudioGen.tile_names()
    from spatialaudiogen_amd.streams import pick_concurrent_streams
    streams.extend(pick_concurrent_streams(NF))
    for n in nets[1:]:                      # the other contexts replay the plan tuned on context 0
        n.inference_ops(*a0)
        for layer, tile, sk, _ in plan:
            n.plan_set(BATCH, layer, names.index(tile) if tile in names else 0, sk)
    tail = None
    if rem:                             # K % G batches left over: one call of a context with that many groups, in the timed region
        tail = SptAudioGen(1, encoders=ENCODERS, separation='unet_mask', groups=rem)
        tail.load_variables(P)
        tail_in = [t[:rem * BATCH] for t in a0]
        tail_out = torch.empty(rem * BATCH, 4800, 3, device='cuda')
        tail.inference_ops(*tail_in, out=tail_out)
        for layer, tile, sk, _ in plan:
            tail.plan_set(BATCH, layer, names.index(tile) if tile in names else 0, sk)
        tail.inference_ops(*tail_in, out=tail_out)

    def region():                       # K steps: n_calls grouped calls rotating over the contexts in flight, then the remainder
        for _ in range(n_calls):
            step()
        if tail is not None:
            with torch.cuda.stream(streams[0] if NF > 1 else torch.cuda.current_stream()):
                tail.inference_ops(*tail_in, out=tail_out)
    torch.cuda.synchronize()
    if args.pmc_child:                  # under rocprofv3 --pmc: a few forwards of the replayed plan on ONE context, nothing else
        for _ in range(3):
            net.inference_ops(*a0, out=outs[0])
        torch.cuda.synchronize()
        return
    # untimed set-up that is NOT a step first (the metric reduction loads the torch kernels it uses - hundreds of milliseconds of an idle
    # GPU on a fresh process), THEN the W warm-up steps, so that they directly precede the timed region: with the reduction between
    # them the chip had idled back to its low clocks when the K timed steps began (first region 4-6 % below the repeats behind it)
    step()
    reduce_metric()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_calls)]
    # (grouped calls: W is rounded UP to whole calls, the same number on every context in flight - a context whose last call lies before the
    #  metric reduction above starts the timed region cold: K = 20, same box: first region 3 092 - 3 110 with one warm-up call, 3 123 - 3 168
    #  with one per context, 3 168 - 3 176 with two, against 3 155 - 3 185 for the regions behind it)
    warm_calls = NF * ((max(args.warmup, 1) + G * NF - 1) // (G * NF))
    for _ in range(warm_calls):
        step()
    torch.cuda.synchronize()
    for e in eval_sums:
        e.zero_()
    counter[0] = 0
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_calls):
        st = streams[counter[0] % NF] if NF > 1 else torch.cuda.current_stream()
        ev[i][0].record(st)
        step()
        ev[i][1].record(st)
    if tail is not None:
        with torch.cuda.stream(streams[0] if NF > 1 else torch.cuda.current_stream()):
            tail.inference_ops(*tail_in, out=tail_out)
    reduce_metric()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    ranks = ranks_report(dist, torch, world, backend, elapsed, steps)
    assert ranks['ranks_seen'] == world, ranks
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if backend != 'gloo' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ns = torch.tensor([float(steps)], dtype=torch.float64, device='cuda' if backend != 'gloo' else 'cpu')
        dist.all_reduce(ns)
        total_steps = int(ns.item())
    else:
        total_steps = steps
    if os.environ.get('BENCH_DEBUG'):
        t00 = ev[0][0]
        print('step timeline (start, end ms):', [(round(t00.elapsed_time(a), 2), round(t00.elapsed_time(b), 2)) for a, b in ev[:8]], file=sys.stderr)
    step_ms = sorted(a.elapsed_time(b) for a, b in ev) or [0.0]

    windows = BATCH * total_steps                       # all ranks
    value = 0.1 * windows / elapsed
    ms_per_step = 1e3 * elapsed / max(steps, 1)

    # ---- extra legs (after the timed region; rank-local): strictly one forward at a time, and host-resident inputs ----
    extra = {}
    if not args.no_repeats and not is_eval and world == 1:
        # the contract's timed region is K steps = a few tens of milliseconds, and boxes / clock ramps move it by several per cent:
        # four MORE regions of the same K steps (same contexts, same batches, same reduction), so that a gain of a few per cent
        # can be told from noise.  The headline fields above are those of the FIRST region only and are not touched.
        reps = [value]
        for _ in range(4):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            region()
            reduce_metric()
            torch.cuda.synchronize()
            reps.append(0.1 * BATCH * steps / (time.perf_counter() - t1))
        extra['headline_repeats'] = {'median': round(float(np.median(reps)), 2), 'min': round(min(reps), 2), 'max': round(max(reps), 2),
                                     'values': [round(r, 2) for r in reps], 'unit': 'ambisonic-s/s',
                                     'note': 'the headline region followed by four more timed regions of %d steps each in the same process '
                                             '(first value = the headline); spread = what one region resolves' % steps}
    if not args.no_extra_legs and not is_eval:
        ks = max(5, min(steps, 20))
        a = batch_inputs(my_batches[0])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(ks):
            net.inference_ops(*a, out=outs[0])          # context 0 on the current stream, nothing else in flight
        torch.cuda.synchronize()
        extra['one_in_flight'] = {'value': round(0.1 * CALLB * world * ks / (time.perf_counter() - t1), 2), 'unit': 'ambisonic-s/s',
                                  'note': 'strictly sequential forward calls on one context (this rank x n_gpus), %d calls of %d batch(es)' % (ks, G)}
        if float_video is not None and not args.float_frames:
            def float_inputs(b):        # the SAME three resident batches the headline cycles, frames as float32
                lo = (b * CALLB) % POOL
                return [float_video[lo:lo + CALLB] if k == 'video' else dev_in[k][lo:lo + CALLB] for k in names_in]
            kf = max(n_calls, 1)        # as long as the headline's region
            for j in range(NF):
                nets[j].inference_ops(*float_inputs(my_batches[j % len(my_batches)]), out=outs[j])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(kf):
                j = i % NF
                ctx = torch.cuda.stream(streams[j]) if NF > 1 else torch.cuda.stream(torch.cuda.current_stream())
                with ctx:
                    nets[j].inference_ops(*float_inputs(my_batches[i % len(my_batches)]), out=outs[j])
            torch.cuda.synchronize()
            extra['float_frames'] = {'value': round(0.1 * CALLB * world * kf / (time.perf_counter() - t1), 2), 'unit': 'ambisonic-s/s',
                                     'note': 'the same forward on float32 frames (the float entry point sagen_forward: the stem runs stem8pool_kernel<MODE 1> on '
                                             'two fp16 planes of x * 2^ka, ka from the exact maximum of the batch - three products per multiply where the uint8 '
                                             'stem needs two, plus the amax / prep passes over 45 MB instead of 16 MB of frames); the same %d resident batches '
                                             'cycled, %d batches in flight, %d steps (compare with headline_repeats, not with the single headline region)' % (len(my_batches), NF, kf)}
        # H2D-inclusive: every batch starts in pinned host memory; copy (on the step's stream) + forward, NF in flight
        # frames travel as the uint8 the JPEG decoder produced (x/255 - 0.5 is applied on the device, bit-identical: the synthetic
        # frames are exact images of uint8 values, checked here); audio (and flow, which is decoded to float) as fp32
        host = []
        for k, t in zip(names_in, a):
            if k == 'video' and t.dtype != torch.uint8:
                u8 = torch.round((t.double() + 0.5) * 255.0).clamp(0, 255).to(torch.uint8)
                if bool(((u8.double() / 255.0 - 0.5).float() == t).all()):
                    t = u8
            host.append(t.cpu().pin_memory())
        # one copy stream shared by the contexts (the link is serial anyway) and two device buffer sets per context: the copy of a
        # context's next batch travels under its current forward (round 5; before, copy and forward of a batch were serial on the
        # step's stream: 2 059 against 2 459 device-resident)
        devb = [[[torch.empty_like(t, device='cuda') for t in host] for _ in range(2)] for _ in range(NF)]
        copy_stream = torch.cuda.Stream()
        ready = [[torch.cuda.Event() for _ in range(2)] for _ in range(NF)]
        consumed = [[torch.cuda.Event() for _ in range(2)] for _ in range(NF)]
        mb = sum(h.numel() * h.element_size() for h in host) / 1e6

        def h2d_loop(n):
            for i in range(n):
                j, par = i % NF, (i // NF) % 2
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(consumed[j][par])          # (never recorded yet: no wait) the forward that last read this set is done
                    for d_, h_ in zip(devb[j][par], host):
                        d_.copy_(h_, non_blocking=True)
                    ready[j][par].record(copy_stream)
                st = streams[j] if NF > 1 else torch.cuda.current_stream()
                with torch.cuda.stream(st):
                    st.wait_event(ready[j][par])
                    nets[j].inference_ops(*devb[j][par], out=outs[j])
                    consumed[j][par].record(st)
        h2d_loop(2 * NF)                                              # (touches every buffer set once)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        h2d_loop(ks)
        torch.cuda.synchronize()
        extra['h2d_inclusive'] = {'value': round(0.1 * CALLB * world * ks / (time.perf_counter() - t1), 2), 'unit': 'ambisonic-s/s',
                                  'note': 'inputs start in pinned host memory (video frames as uint8, normalised on the device): %.1f MB copied per batch on a copy stream into '
                                          'the second of two device buffer sets while the previous batch of the context runs; %d batches in flight, %d steps' % (mb, NF, ks)}

    # ---- roofline of the dominant kernel: per-launch HIP events recorded by the native runtime on the
    #      launch stream (sagen_profile_*), three extra forwards outside the timed region ----
    torch.cuda.synchronize()
    net.profile_enable(BATCH, True)
    agg = {}
    nprof = 3
    for _ in range(nprof):
        net.inference_ops(*a0, out=outs[0])              # context 0 alone on the current stream: unoverlapped launch times
        for k, layer, us, fl in net.profile_report(BATCH):
            a = agg.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1; a[1] += us; a[2] += fl * G          # (the runtime reports the work of ONE group; a grouped launch carries G)
    net.profile_enable(BATCH, False)
    total_us = sum(a[1] for a in agg.values())
    dom = max((k for k in agg if agg[k][2] > 0), key=lambda k: agg[k][1])        # the dominant CONTRACTION (the path is matrix-bound)
    n_l, us_l, fl_l = agg[dom]
    achieved = fl_l / (us_l * 1e-6) / 1e12
    # the largest HBM-bound kernel family beside it (the plane passes): algorithmic bytes are not tracked per launch, so only its share
    hbm_dom = max((k for k in agg if agg[k][2] == 0), key=lambda k: agg[k][1], default=None)
    traffic, traffic_src = None, None
    if rank == 0 and world == 1 and not args.no_pmc:
        traffic, traffic_src = pmc_traffic_in_run(args, net, BATCH, dom)
    if traffic is None:
      try:   # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/)
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            traffic = json.load(f).get(dom)
        if traffic is not None:
            traffic_src = ('profiles/pmc_traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (FETCH_SIZE x2 per the '
                           'gfx950 note), NOT measured in this run' % committed_rev('profiles/pmc_traffic.json'))
      except Exception:
        pass
    step_tflops = cfg['gflop'] * BATCH / (ms_per_step * 1e-3) / 1e3          # throughput-based (steps overlap when NF > 1)
    def is_h2(k):        # three matrix products per fp32 multiply: fp16x2 planes (conv3h, conv3g<..., true, ...>) / the exact uint8 plane
        return k.startswith('conv3h') or k.startswith('stem8pool') or (k.startswith('conv3g') and ',true' in k)

    def is_b3(k):        # the six-product bf16x3 family
        return k.startswith('igemm3') or k.startswith('conv3p') or (k.startswith('conv3g') and ',true' not in k)
    h2, b3 = is_h2(dom), is_b3(dom)
    peak = PEAK_FP16X2_TFLOPS if h2 else (PEAK_BF16X3_TFLOPS if b3 else PEAK_FP32_MFMA_TFLOPS)
    b3_us = sum(a[1] for k, a in agg.items() if is_b3(k))
    h2_us = sum(a[1] for k, a in agg.items() if is_h2(k))
    f32_us = sum(a[1] for k, a in agg.items() if k.startswith('igemm_kernel'))
    roofline = {
        'bound': 'mfma', 'kernel': dom, 'achieved': round(achieved, 2), 'peak': round(peak, 1),
        'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': traffic, 'traffic_source': traffic_src,
        'frac_of_sustained_mfma_rate': round(achieved / (SUSTAINED_BF16X3_TFLOPS * (2.0 if h2 else 1.0)), 4) if (b3 or h2) else None,
        'sustained_note': (SUSTAINED_NOTE.replace('290 TFLOP/s', '580 TFLOP/s (three products per multiply)') if h2 else SUSTAINED_NOTE) if (b3 or h2) else None,
        'peak_basis': ('dense fp16 MFMA peak 2500 TF / 3 products per fp32 multiply (fp16x2 planes, conv3h_kernel); algorithmic fp32 FLOPs' if h2 else
                       'dense bf16 MFMA peak 2500 TF / 6 products per fp32 multiply (bf16x3 kernel); algorithmic fp32 FLOPs'
                       if b3 else 'fp32 MFMA peak (v_mfma_f32_32x32x2_f32)'),
        'achieved_over_fp32_mfma_peak': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
        'launches_per_step': round(n_l / nprof / G, 2), 'batches_per_launch': G, 'avg_launch_us': round(us_l / n_l, 2),
        'gflop_per_launch': round(fl_l / n_l / 1e9, 3),
        'share_of_step_time': round(us_l / total_us, 3),
        'largest_hbm_bound_kernel': None if hbm_dom is None else {'kernel': hbm_dom, 'launches_per_step': agg[hbm_dom][0] // nprof,
                                                                  'us_per_step': round(agg[hbm_dom][1] / nprof, 1), 'share_of_step_time': round(agg[hbm_dom][1] / total_us, 3)},
        'contraction_time_split': {'fp16x2_kernels_us': round(h2_us / nprof, 1), 'bf16x3_kernels_us': round(b3_us / nprof, 1), 'fp32_mfma_kernels_us': round(f32_us / nprof, 1)},
        'whole_step': {'achieved': round(step_tflops, 2), 'frac': round(step_tflops / peak, 4),
                       'achieved_over_fp32_mfma_peak': round(step_tflops / PEAK_FP32_MFMA_TFLOPS, 4),
                       'gflop_per_window_needed_only': cfg['gflop'],
                       'kernel_time_us_per_step': round(total_us / nprof / G, 1)},
    }

    result = {
        'metric': 'ambisonic seconds generated/sec (0.1 s windows, 224x448 video)',
        'value': round(value, 2), 'unit': 'ambisonic-s/s', 'n_gpus': world, 'steps': steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': cfg['scaling'], 'vs_baseline': None,
        'dtype': 'f32 (fp32 in / out / accumulate; products on the 16-bit matrix cores with fp32-equivalent operand splits: two fp16 planes '
                 '(3 products per multiply) for the ResNet trunk\'s 3x3 convs, whose operand ranges batch-norm bounds, three bf16 planes '
                 '(6 products) elsewhere - measured against the fp64 oracle as accurate as the exact fp32 MFMA kernels '
                 '(profiles/r04_accuracy_modes.jsonl; SAGEN_FP32_ONLY=1 selects those; parity bar 1e-4 RMS unchanged)',
        'data': 'synthetic' + (' (pool of %d distinct windows cycled over the %d x %d window set)' % (POOL, EVAL_CLIPS, EVAL_WINDOWS_PER_CLIP) if is_eval else
                               ' (%d distinct resident batches per GPU, cycled)' % NPOOL),
        'config': {'workload': cfg['workload'], 'name': args.config,
                   'windows_per_gpu_per_step': BATCH, 'windows_per_s': round(windows / elapsed, 1),
                   'sharding': ('whole batches of the global window order over ranks (%d batches in total, %d timed on this rank); '
                                '1 metric all-reduce at the end' % (n_batches, steps)) if is_eval else
                               'windows/clips over ranks, no data-path collective; 1 metric all-reduce at the end',
                   'weights': 'random init (Xavier / BN identity), same replica on every rank',
                   'launch_plan': 'autotuned per layer (%d contractions)' % len(plan) if plan else 'shape heuristics',
                   'video_frames': frames_note if 'video' in dev_in else None,
                   'batches_in_flight': NF * G, 'contexts_in_flight': NF, 'batches_per_grouped_launch': G, 'warmup_steps_run': warm_calls * G,
                   'grouped_launch': None if G == 1 else
                   ('every forward call carries %d INDEPENDENT batches of %d windows as one launch per layer (sagen_forward_grouped: the group is a grid '
                    'dimension; own batch-norm statistics / plane scales per batch, outputs bit-identical to one call per batch - '
                    'tests/test_gpu_grouped.py); a step is one batch: %d steps = %d calls%s' % (G, BATCH, steps, n_calls, ' + one call of %d' % rem if rem else ''))},
        'ranks': ranks,
        'step_latency_ms_event': {'median': round(float(np.median(step_ms)), 4), 'p10': round(step_ms[len(step_ms) // 10], 4),
                          'p90': round(step_ms[(9 * len(step_ms)) // 10], 4)},
        'roofline': roofline,
    }
    result.update(extra)
    if is_eval:
        tot = metric[2:].cpu().numpy().reshape(4, 3) / max(windows, 1)
        result['eval_metric_means'] = {k: [round(float(x), 6) for x in tot[i]] for i, k in enumerate(('stft', 'lsd', 'mse', 'snr'))}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sample = {k: v[:BATCH] for k, v in inp.items()}
        cb = cpu_baseline(P, sample, args.cpu_seconds, ENCODERS, BATCH)
        cb['gpu_over_cpu'] = round(value / cb['value'], 1)
        result['cpu_baseline'] = cb
    if rank == 0 and world == 1 and args.config == 'av' and not args.no_other_configs and not args.no_extra_legs:
        torch.cuda.synchronize()
        result.update(other_config_legs(args))
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
