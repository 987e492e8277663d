"""Dev helper (GPU box): run forwards with guard bands around the native workspace, report out-of-bounds writes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spatialaudiogen_amd.model as M
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
G = 64 << 20        # floats of guard on either side (256 MiB)
SENT = 1234.5
bigs = []
orig_empty = torch.empty
def guarded_empty(*a, **k):
    if len(a) == 1 and isinstance(a[0], int) and a[0] > (20 << 20) and k.get('dtype') == torch.float32:
        n = a[0]
        big = orig_empty(n + 2 * G, **k); big.fill_(SENT)
        bigs.append((big, n))
        return big[G:G + n]
    return orig_empty(*a, **k)
M.torch.empty = guarded_empty
encs = [['audio', 'video']] if len(sys.argv) < 2 else [sys.argv[1].split(',')]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
for enc in encs:
    P = init_weights(variable_specs(enc), seed=0, mode='bench')
    inp = synth_inputs(B, enc, seed=1)
    net = M.SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    a = torch.as_tensor(inp['audio']).cuda(); v = torch.as_tensor(inp['video']).cuda() if 'video' in inp else None
    f = torch.as_tensor(inp['flow']).cuda() if 'flow' in inp else None
    def check(tag):
        torch.cuda.synchronize()
        for big, n in bigs:
            lo = big[:G]; hi = big[G + n:]
            for nm, t, base in (('below', lo, -G), ('above', hi, n)):
                bad = (t != SENT).nonzero().flatten()
                if bad.numel():
                    idx = bad.cpu().numpy()
                    print('%s: %d floats overwritten %s the workspace; offsets (floats, relative to workspace start) %d..%d; first values %s' % (
                        tag, idx.size, nm, base + idx.min(), base + idx.max(), t[bad[:4]].cpu().numpy()), flush=True)
                    t[bad] = SENT
    net.inference_ops(a, v, f); check('heuristic forward')
    if os.environ.get('TUNE', '1') == '1':
        net.autotune(a, v, f); check('autotune')
        net.inference_ops(a, v, f); check('tuned forward')
    print('workspace floats:', bigs[-1][1], 'done', enc, flush=True)
