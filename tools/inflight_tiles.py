"""Dev helper (GPU box): headline throughput with NF batches in flight while the stride-1 3x3 trunk convs are pinned to one conv3h tile
(the autotuner times every layer alone on the chip; with several batches in flight the co-resident workgroups of the OTHER batches
change what the best tile is).  usage: NF=3 python tools/inflight_tiles.py"""
import sys, os, time
os.environ.setdefault('SAGEN_ONE_STREAM', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
from spatialaudiogen_amd.model import SptAudioGen
from spatialaudiogen_amd.streams import pick_concurrent_streams
enc = ['audio', 'video']; B = 32
NF = int(os.environ.get('NF', '3'))
P = init_weights(variable_specs(enc), seed=0, mode='bench')
inp = synth_inputs(B, enc, seed=1)
a = torch.as_tensor(inp['audio']).cuda()
v = torch.round((torch.as_tensor(inp['video']).cuda().double() + 0.5) * 255.0).clamp(0, 255).to(torch.uint8)
nets = [SptAudioGen(1, encoders=enc, separation='unet_mask') for _ in range(NF)]
outs = []
for n in nets:
    n.load_variables(P); outs.append(n.inference_ops(a, v))
nets[0].autotune(a, v)
nets[0].save_plan(B, '/tmp/plan_if.json')
streams = pick_concurrent_streams(NF)
names = SptAudioGen.tile_names()
layers = ['video_encoder/conv%d_%d/conv_%d' % (s, u, c) for s in (2, 3, 4, 5) for u in (1, 2) for c in (1, 2) if not (s > 2 and u == 1 and c == 1)]
def run(N=90):
    for i in range(2 * NF):
        with torch.cuda.stream(streams[i % NF]): nets[i % NF].inference_ops(a, v, out=outs[i % NF])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N):
        with torch.cuda.stream(streams[i % NF]): nets[i % NF].inference_ops(a, v, out=outs[i % NF])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return 0.1 * B * N / dt
cands = [None] + [t for t in names if t.startswith('conv3h_kernel')]
for rep in range(2):
    for t in cands:
        for n in nets:
            n.load_plan(B, '/tmp/plan_if.json')
            if t is not None:
                stages = os.environ.get('STAGES', '2345')
                for l in layers:
                    if l[len('video_encoder/conv')] in stages:
                        n.plan_set(B, l, names.index(t), 1)
        try:
            print('%-34s %8.1f ambisonic-s/s' % (t or 'tuned plan', run()), flush=True)
        except Exception as e:
            print('%-34s failed: %s' % (t, str(e)[:80]), flush=True)
