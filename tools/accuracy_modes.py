"""Dev helper (GPU box): end-to-end error of the HIP path against the fp64 oracle, in the default arithmetic (bf16x3
contractions) and with SAGEN_FP32_ONLY=1 (exact fp32 MFMA).  Prints one JSON line per (mode, encoders, seed)."""
import sys, os, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import numpy as np, torch
    from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
    from spatialaudiogen_amd.model import SptAudioGen
    from oracle.np_oracle import SptAudioGenOracle
    mode = 'fp32_only' if os.environ.get('SAGEN_FP32_ONLY') else ('bf16x3' if os.environ.get('SAGEN_NO_H2') else
            'bf16x3 + fp16x2 trunk planes from stage %s' % os.environ.get('SAGEN_P3_FROM_STAGE', '2 (default)'))
    if os.environ.get('SAGEN_NO_DECONV_SCATTER'):
        mode += ', round-4 decoder and trunk (no scatter form, no deconv1 planes, fp32 residuals)'
    elif not os.environ.get('SAGEN_FP32_ONLY') and not os.environ.get('SAGEN_NO_H2'):
        mode += ', round-5 decoder' + (' with the scatter GEMMs forced onto fp16x2 planes (the B >= 16 default)' if os.environ.get('ACC_DECODER_PLANES') else ' (B = 4: scatter GEMMs on fp32 operands, deconv1 on planes)')
    G = int(os.environ.get('ACC_GROUPS', '1'))      # round 6: G batches of 4 per grouped call (B * G >= 128 also moves the audio encoder's conv2 .. conv5 onto fp16x2 planes)
    if G > 1:
        mode += ', %d batches per grouped call: audio encoder conv2..conv5 on fp16x2 planes of the concat buffers, group 0 against the oracle' % G
    also_conv5 = True
    for enc in (['audio'], ['audio', 'video'], ['audio', 'video', 'flow']):
        for seed in (0, 1):
            P = init_weights(variable_specs(enc), seed=seed, mode='test')
            inp = synth_inputs(4, enc, seed=100 + seed)
            ref = SptAudioGenOracle(encoders=enc).inference_ops(inp['audio'], P, video=inp.get('video'), flow=inp.get('flow'))
            net = SptAudioGen(1, encoders=enc, separation='unet_mask', groups=G)
            net.load_variables(P)
            if G > 1:         # group 0 = the oracle's batch; the other groups: other windows (their own statistics)
                more = synth_inputs(4 * (G - 1), enc, seed=900 + seed)
                inp = {k: np.concatenate([inp[k], more[k]], 0) for k in inp}
            out = net.inference_ops(inp['audio'], inp.get('video'), inp.get('flow'))[:4]
            if G > 1:         # the audio encoder's conv2 .. conv5 on planes: what a tuned grouped context runs (untuned, only from batch 128 on)
                tid = SptAudioGen.tile_names().index('conv3g_kernel<128,128,64,64,2,true>')
                for l in range(2, 6):
                    net.plan_set(4, 'audio_encoder/conv%d' % l, tid, 1)
            if os.environ.get('ACC_DECODER_PLANES'):
                net.set_option(4, 'decoder_planes', 1)
                out = net.inference_ops(inp['audio'], inp.get('video'), inp.get('flow'))[:4]
            out = out.cpu().numpy().astype(np.float64)
            err = float(np.sqrt(np.mean((out - ref) ** 2))); rms = float(np.sqrt(np.mean(ref ** 2)))
            trunk = None
            if 'video' in enc:
                orc_t = SptAudioGenOracle(encoders=enc); orc_t.inference_ops(inp['audio'][:4], P, video=None if inp.get('video') is None else inp['video'][:4], flow=None if inp.get('flow') is None else inp['flow'][:4])
                t = net.intermediate(4, 'video_encoder/conv5_2').cpu().numpy().astype(np.float64); r_ = orc_t.ends['video_encoder/conv5_2']       # (group 0's)
                trunk = float(np.sqrt(np.mean((t - r_) ** 2)) / np.sqrt(np.mean(r_ ** 2)))
            print(json.dumps({'mode': mode, 'trunk_conv5_2_rel_err': trunk, 'encoders': '+'.join(e[0].upper() for e in enc), 'seed': seed, 'rms_err': err,
                              'out_rms': rms, 'rel': err / rms, 'max_abs_err': float(np.abs(out - ref).max())}), flush=True)
else:
    for env in ({}, {'ACC_DECODER_PLANES': '1'}, {'ACC_DECODER_PLANES': '1', 'ACC_GROUPS': '32'}, {'SAGEN_NO_DECONV_SCATTER': '1', 'SAGEN_NO_DECONV1_PLANES': '1', 'SAGEN_NO_LEAN_TRUNK': '1'},
                {'SAGEN_NO_H2': '1'}, {'SAGEN_FP32_ONLY': '1'}):
        e = dict(os.environ); e.update(env)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), 'child'], env=e)
