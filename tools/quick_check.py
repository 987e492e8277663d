"""Dev helper: print end-to-end parity numbers and a first timing on the GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
from spatialaudiogen_amd.model import SptAudioGen
from oracle.np_oracle import SptAudioGenOracle

print(torch.cuda.get_device_name(0), os.cpu_count(), 'cpus')
for enc in (['audio'], ['audio', 'video'], ['audio', 'video', 'flow']):
    P = init_weights(variable_specs(enc), seed=0, mode='test')
    inp = synth_inputs(2, enc, seed=1234)
    t = time.time()
    ref = SptAudioGenOracle(encoders=enc).inference_ops(inp['audio'], P, video=inp.get('video'), flow=inp.get('flow'))
    t_or = time.time() - t
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    out = net.inference_ops(inp['audio'], inp.get('video'), inp.get('flow')).cpu().numpy()
    err = np.sqrt(np.mean((out - ref) ** 2)); r = np.sqrt(np.mean(ref ** 2))
    print(enc, 'oracle %.1fs' % t_or, 'rms err %.3g  out rms %.3g  rel %.3g  max|out| %.3g' % (err, r, err / r, np.abs(out).max()))

for enc, B in ((['audio', 'video'], 32), (['audio'], 32), (['audio', 'video', 'flow'], 32)):
    P = init_weights(variable_specs(enc), seed=0, mode='bench')
    inp = synth_inputs(B, enc, seed=1234)
    net = SptAudioGen(1, encoders=enc, separation='unet_mask')
    net.load_variables(P)
    a = torch.as_tensor(inp['audio']).cuda(); v = torch.as_tensor(inp['video']).cuda() if 'video' in inp else None
    f = torch.as_tensor(inp['flow']).cuda() if 'flow' in inp else None
    out = torch.empty(B, 4800, 3, device='cuda')
    for _ in range(3):
        net.inference_ops(a, v, f, out=out)
    torch.cuda.synchronize()
    t = time.time(); n = 10
    for _ in range(n):
        net.inference_ops(a, v, f, out=out)
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    print(enc, 'B=%d  %.3f ms/batch  %.0f windows/s  %.1f ambisonic-s/s' % (B, dt * 1e3, B / dt, 0.1 * B / dt))
