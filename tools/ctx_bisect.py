"""Dev helper (GPU box): which single tuned layer choice makes two concurrently running contexts disagree with their own
sequential result?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
from spatialaudiogen_amd.model import SptAudioGen
enc = ['audio', 'video']; B = 32; K = 2; iters = 40
P = init_weights(variable_specs(enc), seed=0, mode='bench')
inp = synth_inputs(B, enc, seed=1)
a = torch.as_tensor(inp['audio']).cuda(); v = torch.as_tensor(inp['video']).cuda()
tuner = SptAudioGen(1, encoders=enc, separation='unet_mask'); tuner.load_variables(P)
plan = tuner.autotune(a, v)
names = SptAudioGen.tile_names()
streams = [torch.cuda.Stream() for _ in range(K)]
def torture(rows):
    nets = [SptAudioGen(1, encoders=enc, separation='unet_mask') for _ in range(K)]
    for n in nets:
        n.load_variables(P); n.inference_ops(a, v)
        for layer, tile, sk, us in rows:
            n.plan_set(B, layer, names.index(tile) if tile in names else 0, sk)
    seq = [n.inference_ops(a, v).clone() for n in nets]
    torch.cuda.synchronize()
    outs = [torch.empty_like(seq[0]) for _ in range(K)]
    for i in range(iters):
        with torch.cuda.stream(streams[i % K]): nets[i % K].inference_ops(a, v, out=outs[i % K])
    torch.cuda.synchronize()
    return max(float((outs[j] - seq[j]).abs().max()) for j in range(K))
print('all tuned rows:', torture(plan), flush=True)
print('no rows (heuristics):', torture([]), flush=True)
for row in plan:
    d = torture([row])
    if d > 0: print('BAD  %-46s %-38s sk=%d  -> max diff %.3g' % (row[0], row[1], row[2], d), flush=True)
print('bisect done')
