#!/usr/bin/env python
"""Headline benchmark: ambisonic seconds generated per second (0.1 s windows, 224x448 video).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
            --master-port P bench.py --gpus N --steps K --warmup W)

One "step" = one pass of the hot path (sagen_forward: STFT -> audio + ResNet18 video encoders ->
U-Net mask decoder -> iSTFT -> ambisonic mix) over one batch of 32 synthetic 0.1 s windows per GPU
(BASELINE.json configs[1]).  Inputs are resident in HBM before the timed region.  Windows shard over
ranks with no data-path collective (weak scaling); the only collective is the eval-style metric
all-reduce (RCCL) issued once at the end of the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 32
ENCODERS = ['audio', 'video']
# needed-only algorithmic work per window, A+V (BASELINE.md 2 / SURVEY.md 8d)
GFLOP_PER_WINDOW = 8.41
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0         # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16: 32 cycles / SIMD)
# igemm3_kernel evaluates every fp32 product as 6 bf16 products (bf16x3 operand split, fp32 accumulate): its matrix roof
# in ALGORITHMIC fp32 FLOP/s is the bf16 peak / 6
PEAK_BF16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--in-flight', type=int, default=2,
                    help='batches in flight per GPU: steps rotate over this many native contexts, each on its own stream, the streams '
                         'probed to really run concurrently (DESIGN.md 6.1); 1 = strictly one forward at a time')
    ap.add_argument('--no-autotune', action='store_true', help='use shape heuristics instead of the timed per-layer plan')
    ap.add_argument('--plan-file', default=None, help='replay this saved launch plan if it exists, else autotune and save it')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='budget of the CPU baseline leg')
    return ap.parse_args()


def cpu_baseline(P, inputs, budget_s):
    """The reference-equivalent CPU path (oracle/torch_ref.py, torch-CPU/oneDNN fp32, all host
    threads) on the same synthetic batch; TF 1.4 cannot be installed offline."""
    import torch
    from oracle.torch_ref import TorchRef
    threads = torch.get_num_threads()
    ref = TorchRef(P, ENCODERS, dtype=torch.float32)
    a, v = inputs['audio'], inputs['video']
    t0 = time.time()
    ref.forward(a, v)                                  # warm-up (oneDNN primitive creation)
    warm = time.time() - t0
    times = []
    t_start = time.time()
    while len(times) < 2 or (time.time() - t_start < budget_s and len(times) < 20):
        t0 = time.time()
        ref.forward(a, v)
        times.append(time.time() - t0)
    med = float(np.median(times))
    return {
        'value': round(0.1 * BATCH / med, 3), 'unit': 'ambisonic-s/s', 'cores': threads, 'kind': 'port',
        'sample': '%d timed batches of %d windows (same synthetic A+V workload, fp32), median %.3f s/batch, '
                  'warm-up %.2f s; torch-CPU/oneDNN stand-in for the TF1 CPU path (TF1 unavailable offline)'
                  % (len(times), BATCH, med, warm),
    }


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        # convenience: re-launch ourselves under torchrun, one rank per GPU
        import subprocess
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', os.environ.get('MASTER_PORT', '29533'),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    if args.in_flight > 1:
        # several batches in flight: each context stays on its caller's stream (the other batch is the overlap)
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
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # RCCL ('nccl') is the backend; SAGEN_DIST_BACKEND=gloo is a test hook for running several ranks on ONE GPU
        dist.init_process_group(os.environ.get('SAGEN_DIST_BACKEND', 'nccl'), rank=rank, world_size=world)

    P = init_weights(variable_specs(ENCODERS), seed=0, mode='bench')           # same replica on every rank
    inp = synth_inputs(BATCH, ENCODERS, seed=1234 + rank)                      # each rank owns its windows
    # Steps are independent batches, so NF of them are kept in flight: step i runs on native context i % NF and stream
    # i % NF (each context has its own workspace and its own second stream).  Every step is still one full forward of
    # one batch; the kernels of neighbouring steps fill each other's launch tails.  Outputs are bit-identical to
    # strictly sequential execution (tests/test_gpu_streams.py, tools/two_in_flight.py).
    NF = max(1, args.in_flight)
    nets = [SptAudioGen(1, encoders=ENCODERS, separation='unet_mask') for _ in range(NF)]
    for n in nets:
        n.load_variables(P)
    net = nets[0]
    streams = []                        # created after the contexts (below): ROCm maps HIP streams to hardware queues in creation order
    audio = torch.as_tensor(inp['audio']).cuda()
    video = torch.as_tensor(inp['video']).cuda()
    outs = [torch.empty(BATCH, 4800, 3, device='cuda') for _ in range(NF)]
    out = outs[0]
    metric = torch.zeros(4, dtype=torch.float64, device='cuda')
    counter = [0]

    def step():
        j = counter[0] % NF
        counter[0] += 1
        if NF == 1:
            net.inference_ops(audio, video, out=out)
        else:
            with torch.cuda.stream(streams[j]):
                nets[j].inference_ops(audio, video, out=outs[j])
        return j

    def barrier():
        if world > 1:
            dist.barrier()

    def reduce_metric():
        # eval-style metric reduction: per-rank sums + count, one all-reduce over RCCL
        if NF > 1:                          # the reduction (current stream) consumes what the step streams produced
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
        metric[0] = sum((o.double() ** 2).sum() for o in outs) / NF
        metric[1] = float(BATCH)
        if world > 1:
            dist.all_reduce(metric)

    # untimed set-up: per-layer (tile, split-K) autotune on this rank's own batch, then W warm-up steps
    plan = []
    if args.plan_file and os.path.exists(args.plan_file):
        net.inference_ops(audio, video, out=out)
        net.load_plan(BATCH, args.plan_file)
        plan = net.plan(BATCH)
    elif not args.no_autotune:
        plan = net.autotune(audio, video)
        if args.plan_file and rank == 0:
            net.save_plan(BATCH, args.plan_file)
    names = SptAudioGen.tile_names()
    from spatialaudiogen_amd.streams import pick_concurrent_streams
    streams.extend(pick_concurrent_streams(NF))
    for n in nets[1:]:                      # the other contexts replay the plan tuned on context 0
        n.inference_ops(audio, video)
        for layer, tile, sk, _ in plan:
            n.plan_set(BATCH, layer, names.index(tile) if tile in names else 0, sk)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    reduce_metric()                     # also loads the torch kernels it uses before the timed region
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        st = streams[counter[0] % NF] if NF > 1 else torch.cuda.current_stream()
        ev[i][0].record(st)
        step()
        ev[i][1].record(st)
    reduce_metric()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if os.environ.get('BENCH_DEBUG'):
        t00 = ev[0][0]
        print('step timeline (start, end ms):', [(round(t00.elapsed_time(a), 2), round(t00.elapsed_time(b), 2)) for a, b in ev[:8]], file=sys.stderr)
    step_ms = sorted(a.elapsed_time(b) for a, b in ev)

    windows = world * BATCH * args.steps
    value = 0.1 * windows / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel: per-launch HIP events recorded by the native runtime on the
    #      launch stream (sagen_profile_*), three extra forwards outside the timed region ----
    torch.cuda.synchronize()
    net.profile_enable(BATCH, True)
    agg = {}
    nprof = 3
    for _ in range(nprof):
        net.inference_ops(audio, video, out=out)          # context 0 alone on the current stream: unoverlapped launch times
        for k, layer, us, fl in net.profile_report(BATCH):
            a = agg.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1; a[1] += us; a[2] += fl
    net.profile_enable(BATCH, False)
    total_us = sum(a[1] for a in agg.values())
    dom = max(agg, key=lambda k: agg[k][1])
    n_l, us_l, fl_l = agg[dom]
    achieved = fl_l / (us_l * 1e-6) / 1e12
    traffic = None
    try:     # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/)
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            traffic = json.load(f).get(dom)
    except Exception:
        pass
    step_tflops = GFLOP_PER_WINDOW * BATCH / (ms_per_step * 1e-3) / 1e3          # throughput-based (steps overlap when NF > 1)
    b3 = dom.startswith('igemm3') or dom.startswith('conv3p')   # igemm3 / igemm3dw / igemm3s2 / conv3p kernels: the bf16x3 family
    peak = PEAK_BF16X3_TFLOPS if b3 else PEAK_FP32_MFMA_TFLOPS
    b3_us = sum(a[1] for k, a in agg.items() if k.startswith('igemm3') or k.startswith('conv3p'))
    f32_us = sum(a[1] for k, a in agg.items() if k.startswith('igemm_kernel'))
    roofline = {
        'bound': 'mfma', 'kernel': dom, 'achieved': round(achieved, 2), 'peak': round(peak, 1),
        'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': traffic,
        'peak_basis': ('dense bf16 MFMA peak 2500 TF / 6 products per fp32 multiply (bf16x3 kernel); algorithmic fp32 FLOPs'
                       if b3 else 'fp32 MFMA peak (v_mfma_f32_32x32x2_f32)'),
        'achieved_over_fp32_mfma_peak': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
        'launches_per_step': n_l // nprof, 'avg_launch_us': round(us_l / n_l, 2),
        'gflop_per_launch': round(fl_l / n_l / 1e9, 3),
        'share_of_step_time': round(us_l / total_us, 3),
        'contraction_time_split': {'bf16x3_kernels_us': round(b3_us / nprof, 1), 'fp32_mfma_kernels_us': round(f32_us / nprof, 1)},
        'whole_step': {'achieved': round(step_tflops, 2), 'frac': round(step_tflops / peak, 4),
                       'achieved_over_fp32_mfma_peak': round(step_tflops / PEAK_FP32_MFMA_TFLOPS, 4),
                       'gflop_per_window_needed_only': GFLOP_PER_WINDOW,
                       'kernel_time_us_per_step': round(total_us / nprof, 1)},
    }

    result = {
        'metric': 'ambisonic seconds generated/sec (0.1 s windows, 224x448 video)',
        'value': round(value, 2), 'unit': 'ambisonic-s/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 (products on the bf16 matrix cores as a 3-way bf16 operand split, 6 products per multiply, fp32 accumulate '
                 '- fp32-equivalent, parity bar 1e-4 RMS unchanged; SAGEN_FP32_ONLY=1 selects the exact fp32 MFMA kernels)',
        'data': 'synthetic',
        'config': {'workload': 'configs[1]: audio+video encoders (no flow), 224x448@10fps + 48 kHz mono, '
                               'batch 32 x 0.1 s windows per GPU, FREQ_MASK separation, 32 tracks',
                   'windows_per_gpu_per_step': BATCH, 'windows_per_s': round(windows / elapsed, 1),
                   'sharding': 'windows/clips over ranks, no data-path collective; 1 metric all-reduce at the end',
                   'weights': 'random init (Xavier / BN identity), same replica on every rank',
                   'launch_plan': 'autotuned per layer (%d contractions)' % len(plan) if plan else 'shape heuristics',
                   'batches_in_flight': NF},
        'step_latency_ms_event': {'median': round(float(np.median(step_ms)), 4), 'p10': round(step_ms[len(step_ms) // 10], 4),
                          'p90': round(step_ms[(9 * len(step_ms)) // 10], 4)},
        'roofline': roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(P, inp, args.cpu_seconds)
        cb['gpu_over_cpu'] = round(value / cb['value'], 1)
        result['cpu_baseline'] = cb
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
