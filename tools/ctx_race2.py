"""Dev helper (GPU box): concurrency torture - N contexts alternate on N streams for many iterations; every context must
reproduce its own sequential output."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
from spatialaudiogen_amd.model import SptAudioGen
enc = sys.argv[1].split(','); B = int(sys.argv[2]); K = int(sys.argv[3]); tune = sys.argv[4] == '1'; iters = int(sys.argv[5])
P = init_weights(variable_specs(enc), seed=0, mode='bench')
inp = synth_inputs(B, enc, seed=1)
a = torch.as_tensor(inp['audio']).cuda(); v = torch.as_tensor(inp['video']).cuda() if 'video' in inp else None
nets = [SptAudioGen(1, encoders=enc, separation='unet_mask') for _ in range(K)]
for n in nets: n.load_variables(P); n.inference_ops(a, v)
if tune:
    for n in nets: n.autotune(a, v)
seq = [n.inference_ops(a, v).clone() for n in nets]
torch.cuda.synchronize()
extra = [torch.cuda.Stream() for _ in range(int(os.environ.get('EXTRA_STREAMS', '0')))]      # idle streams
streams = [torch.cuda.Stream() for _ in range(K)]
if os.environ.get('SAME_STREAM'): streams = [streams[0]] * K
outs = [torch.empty_like(seq[0]) for _ in range(K)]
names = ['mag', 'audio_encoder/conv1', 'audio_encoder/conv5', 'bottleneck', 'separation/deconv1']
refs = [{nm: n.intermediate(B, nm) for nm in names} for n in nets]
for i in range(iters):
    with torch.cuda.stream(streams[i % K]): nets[i % K].inference_ops(a, v, out=outs[i % K])
torch.cuda.synchronize()
print(enc, 'B', B, 'contexts', K, 'tuned', tune, 'iters', iters)
for j in range(K):
    print('  ctx%d out diff %.3g;' % (j, float((outs[j] - seq[j]).abs().max())), {nm: '%.3g' % float((nets[j].intermediate(B, nm) - refs[j][nm]).abs().max()) for nm in names})
G4 = 1 << 32
for j, n in enumerate(nets):
    ws = n.context_for(B).workspace
    base = ws.data_ptr(); size = ws.numel() * 4
    print('  ctx%d workspace [%#x, %#x)  size %.2f GB  crosses 4GiB boundary: %s  base mod 4GiB = %#x' % (j, base, base + size, size / 1e9, (base // G4) != ((base + size - 1) // G4), base % G4))
print('  input a at %#x, v at %#x (%.1f MB)' % (a.data_ptr(), v.data_ptr() if v is not None else 0, (v.numel() * 4 / 1e6) if v is not None else 0))
# --- anatomy of the corruption in 'mag' [B,127,1024] ---
for j, n in enumerate(nets):
    m = n.intermediate(B, 'mag').reshape(B, 127, 1024); r = refs[j]['mag'].reshape(B, 127, 1024)
    bad = (m != r)
    if not bad.any(): continue
    bb = bad.any(dim=2)                          # [B,127] frames with any wrong bin
    idx = bb.nonzero().cpu().numpy()
    print('  ctx%d: %d wrong frames of %d; batch rows %s; frames %s' % (j, len(idx), B * 127, sorted(set(idx[:, 0]))[:12], sorted(set(idx[:, 1]))[:16]))
    b0, f0 = idx[0]
    print('    frame (b=%d,f=%d): wrong bins %d/1024; got[:4] %s  want[:4] %s' % (b0, f0, int(bad[b0, f0].sum()), m[b0, f0, :4].cpu().numpy(), r[b0, f0, :4].cpu().numpy()))
    # is the wrong frame equal to some other reference frame?
    d = (r.reshape(-1, 1024) - m[b0, f0]).abs().amax(dim=1)
    k = int(d.argmin()); print('    closest reference frame: (b=%d,f=%d) max diff %.3g' % (k // 127, k % 127, float(d[k])))
    break
