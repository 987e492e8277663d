"""Diagnostic (GPU box): duration of the stage-2 conv3h launches against the batch size = against the number of 256x64 tiles per CU slot.
784 tiles (B = 32) on 512 slots (256 CUs x 2 workgroups) is 1.53 rounds; this prints the time per launch and per tile for other counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spatialaudiogen_amd.model import SptAudioGen
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs

enc = ['audio', 'video']
P = init_weights(variable_specs(enc), seed=1, mode='test')
net = SptAudioGen(1, encoders=enc, separation='unet_mask')
net.load_variables(P)
for B in [int(a) for a in sys.argv[1:]] or [10, 20, 21, 30, 31, 32, 41, 42]:
    inp = synth_inputs(B, enc, seed=3)
    u8 = np.round((inp['video'].astype(np.float64) + 0.5) * 255.0).astype(np.uint8)
    for _ in range(3):
        net.inference_ops(inp['audio'], u8)
    net.profile_enable(B, True)
    acc = {}
    for _ in range(5):
        net.inference_ops(inp['audio'], u8)
        for k, layer, us, fl in net.profile_report(B):
            if layer.startswith('video_encoder/conv2_') and k.startswith('conv3'):
                acc.setdefault((layer, k), []).append(us)
    net.profile_enable(B, False)
    tiles = (B * 56 * 113 + 255) // 256
    t = [np.median(v) for v in acc.values()]
    ks = sorted({k for (_, k) in acc})
    print('B=%3d tiles(256 rows)=%4d per-slot=%.2f  conv us: %s  mean %.1f  us per 256 tiles %.2f   %s' % (
        B, tiles, tiles / 512., ' '.join('%.1f' % x for x in t), np.mean(t), np.mean(t) / tiles * 256, ks), flush=True)
