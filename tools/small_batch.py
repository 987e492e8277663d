"""Dev helper (GPU box): forward wall time vs summed kernel time at small batch (deploy.py runs 10 windows per call)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
from spatialaudiogen_amd.model import SptAudioGen
for enc in (['audio'], ['audio', 'video'], ['audio', 'video', 'flow']):
    for B in (10, 32):
        P = init_weights(variable_specs(enc), seed=0, mode='bench')
        inp = synth_inputs(B, enc, seed=1)
        net = SptAudioGen(1, encoders=enc, separation='unet_mask')
        net.load_variables(P)
        a = torch.as_tensor(inp['audio']).cuda(); v = torch.as_tensor(inp['video']).cuda() if 'video' in inp else None
        fl = torch.as_tensor(inp['flow']).cuda() if 'flow' in inp else None
        out = net.inference_ops(a, v, fl)
        if os.environ.get('TUNE', '1') == '1': net.autotune(a, v, fl)
        for _ in range(5): net.inference_ops(a, v, fl, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); N = 50
        for _ in range(N): net.inference_ops(a, v, fl, out=out)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / N * 1e6
        t0 = time.perf_counter()
        for _ in range(N): net.inference_ops(a, v, fl, out=out)
        host = (time.perf_counter() - t0) / N * 1e6          # host-side submit time only
        torch.cuda.synchronize()
        net.profile_enable(B, True); net.inference_ops(a, v, fl, out=out); rows = net.profile_report(B); net.profile_enable(B, False)
        ksum = sum(r[2] for r in rows)
        print('%-12s B=%2d  wall %7.1f us  host submit %7.1f us  kernel sum %7.1f us  launches %d  -> %.0f windows/s' % ('+'.join(enc), B, wall, host, ksum, len(rows), B / wall * 1e6), flush=True)
