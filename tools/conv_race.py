"""Dev helper (GPU box): (1) is a bf16x3 / fp32 conv disturbed by a bf16x3 conv on another stream?  (2) does the two-stream
forward reproduce the one-stream forward at B=32 (A+V and A+V+F)?"""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if len(sys.argv) > 1 and sys.argv[1] == 'fwd':
    from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
    from spatialaudiogen_amd.model import SptAudioGen
    enc = sys.argv[2].split(','); B = 32
    P = init_weights(variable_specs(enc), seed=0, mode='bench'); inp = synth_inputs(B, enc, seed=1)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask'); net.load_variables(P)
    a = torch.as_tensor(inp['audio']).cuda(); v = torch.as_tensor(inp['video']).cuda(); f = torch.as_tensor(inp['flow']).cuda() if 'flow' in inp else None
    outs = [net.inference_ops(a, v, f).clone() for _ in range(12)]; torch.cuda.synchronize()
    torch.save(outs[0].cpu(), '/tmp/out_%s_%s.pt' % ('-'.join(enc), 'one' if os.environ.get('SAGEN_ONE_STREAM') else 'two'))
    print('%s %s-stream: 12 repeated forwards identical: %s' % (enc, 'one' if os.environ.get('SAGEN_ONE_STREAM') else 'two', all(torch.equal(outs[0], o) for o in outs)))
    sys.exit(0)
from spatialaudiogen_amd import ops
x = torch.randn(32, 28, 56, 128, device='cuda'); w = torch.randn(3, 3, 128, 128, device='cuda') * 0.05
x2 = torch.randn(32, 56, 112, 64, device='cuda'); w2 = torch.randn(3, 3, 64, 64, device='cuda') * 0.05
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ref, rs = ops.conv_2d(x, w, 1, 'SAME', return_bn_stats=True); torch.cuda.synchronize()
worst = 0.0; ws = 0.0
for _ in range(30):
    with torch.cuda.stream(sb):
        for _ in range(3): ops.conv_2d(x2, w2, 1, 'SAME', return_bn_stats=True)
    with torch.cuda.stream(sa):
        y, st = ops.conv_2d(x, w, 1, 'SAME', return_bn_stats=True)
    torch.cuda.synchronize()
    worst = max(worst, float((y - ref).abs().max())); ws = max(ws, float((st - rs).abs().max()))
print('victim conv (tile %s) vs a concurrent bf16x3 conv: max |y - solo| = %.3g, stats diff %.3g' % (os.environ.get('SAGEN_FORCE_TILE', 'default'), worst, ws))
