"""Dev helper (GPU box): does a workspace that straddles a 4 GiB address boundary break (a) a sequential forward, (b) two
concurrent contexts?  The workspace is carved out of one big tensor at a chosen offset."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spatialaudiogen_amd.model as M
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
enc = ['audio', 'video']; B = 32
G4 = 1 << 32
pool = torch.empty(3 * G4 // 4, dtype=torch.float32, device='cuda')          # 12 GiB arena
pbase = pool.data_ptr()
first_boundary = (pbase + G4 - 1) // G4 * G4
placements = []          # byte addresses where the next workspaces should start
orig_empty = torch.empty
def placed_empty(*a, **k):
    if len(a) == 1 and isinstance(a[0], int) and a[0] > (20 << 20) and k.get('dtype') == torch.float32 and placements:
        addr = placements.pop(0); n = a[0]
        off = (addr - pbase) // 4
        return pool[off:off + n]
    return orig_empty(*a, **k)
M.torch.empty = placed_empty
P = init_weights(variable_specs(enc), seed=0, mode='bench')
inp = synth_inputs(B, enc, seed=1)
a = torch.as_tensor(inp['audio']).cuda(); v = torch.as_tensor(inp['video']).cuda()
def make(addr):
    placements.append(addr)
    n = M.SptAudioGen(1, encoders=enc, separation='unet_mask'); n.load_variables(P); n.inference_ops(a, v); return n
ws_bytes = 1200 << 20
mode = sys.argv[1]
if mode == 'straddle':      # ctx0 straddles the boundary (boundary 600 MiB into it), ctx1 well inside the next 4 GiB window
    addrs = [first_boundary - (600 << 20), first_boundary + (1500 << 20)]
else:                        # neither straddles
    addrs = [first_boundary + (100 << 20), first_boundary + (1500 << 20)]
nets = [make(x) for x in addrs]
for j, n in enumerate(nets):
    ws = n.context_for(B).workspace; b0 = ws.data_ptr(); sz = ws.numel() * 4
    print('ctx%d workspace [%#x, %#x) crosses 4GiB boundary: %s' % (j, b0, b0 + sz, (b0 // G4) != ((b0 + sz - 1) // G4)))
ref = M.SptAudioGen(1, encoders=enc, separation='unet_mask'); ref.load_variables(P)
r = ref.inference_ops(a, v).clone()
seq = [n.inference_ops(a, v).clone() for n in nets]; torch.cuda.synchronize()
print('sequential vs an ordinary context:', [float((s - r).abs().max()) for s in seq])
streams = [torch.cuda.Stream() for _ in nets]; outs = [torch.empty_like(r) for _ in nets]
for i in range(60):
    with torch.cuda.stream(streams[i % 2]): nets[i % 2].inference_ops(a, v, out=outs[i % 2])
torch.cuda.synchronize()
print('concurrent vs sequential:', [float((outs[j] - seq[j]).abs().max()) for j in range(2)])
