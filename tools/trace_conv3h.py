"""Dev helper (GPU box): life of every workgroup of ONE conv3h launch inside the forward (SAGEN_LIB = tools/build_ab/libsagen_trace.so,
built by tools/build_trace_lib.sh): entry -> K loop entered -> K loop left -> epilogue done, in core-clock cycles, plus the slot timeline
per CU (how long a CU slot sits between one workgroup's end and the next one's entry).

    SAGEN_ONE_STREAM=1 SAGEN_LIB=$PWD/tools/build_ab/libsagen_trace.so python tools/trace_conv3h.py [nth launch ...]
"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatialaudiogen_amd import _lib
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
from spatialaudiogen_amd.model import SptAudioGen

enc = ['audio', 'video']
B = int(os.environ.get('B', '32'))
P = init_weights(variable_specs(enc), seed=0, mode='bench')
inp = synth_inputs(B, enc, seed=1234)
GROUPS = int(os.environ.get('NGROUPS', '1'))       # (NGROUPS: bash ignores assignments to GROUPS) grouped launch: GROUPS batches of B per forward call (round 6)
inp = synth_inputs(B * GROUPS, enc, seed=1234)
net = SptAudioGen(1, encoders=enc, separation='unet_mask', groups=GROUPS)
net.load_variables(P)
a = torch.as_tensor(inp['audio']).cuda()
v = torch.round((torch.as_tensor(inp['video']).cuda().double() + 0.5) * 255.0).clamp(0, 255).to(torch.uint8)
for _ in range(3): net.inference_ops(a, v)
if os.environ.get('TUNE', '1') == '1': net.autotune(a, v)
for item in filter(None, os.environ.get('FORCE', '').split(',')):
    tid_, sub = item.split(':')
    for name in variable_specs(enc):
        if name.endswith('/weights') and sub in name: net.plan_set(B, name[:-len('/weights')], int(tid_), int(os.environ.get('FORCE_SK', '1')))
for _ in range(2): net.inference_ops(a, v)
L = _lib.lib()
L.sagen_debug_trace_conv3h.argtypes = [C.c_void_p, C.c_int]
L.sagen_debug_trace_conv3h.restype = None
NWG = 8192
for nth in [int(x) for x in sys.argv[1:]] or [0, 5]:
    buf = torch.zeros(NWG * 8 + 2 * 64 * 8, dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    L.sagen_debug_trace_conv3h(C.c_void_p(buf.data_ptr()), nth)
    net.inference_ops(a, v)
    torch.cuda.synchronize()
    raw = buf.cpu().numpy()
    if os.environ.get('DUMP'): np.save(os.path.join(os.environ['DUMP'], 'trace_launch%d.npy' % nth), raw)
    gt = raw[NWG * 8:].reshape(2, 64, 8)
    t = raw[:NWG * 8].reshape(NWG, 8)
    live = t[:, 3] != 0
    t = t[live]
    n = len(t)
    if n == 0:
        print('launch %d: nothing traced' % nth); continue
    t0, t1, t2, t3 = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    hw, xcc, r0, r1 = t[:, 4], t[:, 5] & 0xf, t[:, 6], t[:, 7]
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    pro, loop, epi, life = t1 - t0, t2 - t1, t3 - t2, t3 - t0
    q = lambda x: '%7d %7d %7d' % tuple(np.percentile(x, [10, 50, 90]))
    print('launch %d: %d workgroups on %d CUs; kernel span %.1f us (100 MHz clock), core clock / 100 MHz = %.2f' % (
        nth, n, len(set(cuid.tolist())), (r1.max() - r0.min()) / 100.0, np.median(life / np.maximum(r1 - r0, 1))))
    print('   cycles p10/p50/p90: prologue %s | K loop %s | epilogue %s | life %s' % (q(pro), q(loop), q(epi), q(life)))
    first = r0 < np.percentile(r0, 40)
    print('   first round (%d): prologue %s | K loop %s | epilogue %s' % (first.sum(), q(pro[first]), q(loop[first]), q(epi[first])))
    print('   later        (%d): prologue %s | K loop %s | epilogue %s' % ((~first).sum(), q(pro[~first]), q(loop[~first]), q(epi[~first])))
    # slot timeline per CU on the 100 MHz clock: workgroups of one CU sorted by entry
    gaps, conc = [], []
    for c in set(cuid.tolist()):
        m = cuid == c
        s0, s1 = r0[m], r1[m]
        o = np.argsort(s0)
        s0, s1 = s0[o], s1[o]
        # concurrency: average number of resident workgroups over the CU's busy span
        span = s1.max() - s0.min()
        conc.append((s1 - s0).sum() / max(span, 1))
        gaps.append((span, len(s0)))
    print('   per CU: workgroups %.2f mean (min %d max %d); resident workgroups averaged over the CU busy span %.2f; CU busy span p50 %.1f us' % (
        np.mean([g[1] for g in gaps]), min(g[1] for g in gaps), max(g[1] for g in gaps), np.mean(conc), np.median([g[0] for g in gaps]) / 100.0))
    # time from kernel start (first entry anywhere) to each workgroup's entry / end
    base = r0.min()
    print('   entry time after kernel start, us: p10/p50/p90/max %s %.1f;  end: %s %.1f' % (
        ' '.join('%.1f' % x for x in np.percentile((r0 - base) / 100.0, [10, 50, 90])), (r0.max() - base) / 100.0,
        ' '.join('%.1f' % x for x in np.percentile((r1 - base) / 100.0, [10, 50, 90])), (r1.max() - base) / 100.0))
    for w, nm in ((0, 'early workgroup'), (1, 'late workgroup')):
        g = gt[w]
        ng = int((g[:, 0] != 0).sum())
        if ng < 3: continue
        g = g[:ng].astype(np.int64)
        ph = np.stack([g[:, 1] - g[:, 0], g[:, 2] - g[:, 1], g[:, 3] - g[:, 2], g[:, 4] - g[:, 3]], 1)
        per = np.diff(g[:, 0])
        print('   %s: %d groups, period p50 %d cycles; phases p50 [vmcnt wait, barrier, setup + first fragments, MFMA run] = %s; groups 1..4: %s' % (
            nm, ng, int(np.median(per)), np.median(ph, 0).astype(int).tolist(), ph[1:5].tolist()))
