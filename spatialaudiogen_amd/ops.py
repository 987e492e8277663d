"""Op-level Python surface over the C ABI — the counterparts of the reference's layer wrappers
(pyutils/tflib/wrappers/core.py: conv_2d :156-220, deconv_2d :96-153, fully_connected :43-93) and
of myutils.stft / istft (myutils.py:119-211).  Tensors are torch CUDA fp32, NHWC like the TF graph.
torch is only the owner of device memory and streams here; all arithmetic is in libsagen_hip.so.
"""
import ctypes as C
import torch

from . import _lib
from ._lib import check


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _twin():
    _lib.lib()
    return _lib.IS_CPU_TWIN


def _stream():
    if _twin():          # libsagen_cpu.so (SAGEN_LIB): host pointers, no stream
        return None
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name):
    if isinstance(t, torch.Tensor) and t.dtype == torch.float64 and name == 'stats':
        return t
    if not (isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.is_cuda != _twin()):
        raise TypeError('%s must be a %s float32 tensor' % (name, 'host (the CPU twin is loaded)' if _twin() else 'CUDA'))
    return t.contiguous()


def _scratch(nbytes, dev):
    return torch.empty((int(nbytes) + 255) // 256 * 64, dtype=torch.float32, device=dev)


def stft_mag(audio, f0, f1, c0=None, c1=None):
    """myutils.stft (myutils.py:119-147) + crop + tf.abs (model.py:166-178).
    audio [B, n]; returns (mag [B,f1-f0,1024], spec [B,c1-c0,513,2] or None)."""
    audio = _f32(audio, 'audio')
    B, n = audio.shape
    mag = torch.empty(B, f1 - f0, 1024, dtype=torch.float32, device=audio.device)
    spec = None
    if c0 is not None:
        spec = torch.empty(B, c1 - c0, 513, 2, dtype=torch.float32, device=audio.device)
    check(_lib.lib().sagen_stft_mag(_ptr(audio), B, n, f0, f1, _ptr(mag), c0 or 0, c1 or 0, _ptr(spec), _stream()))
    return mag, spec


def conv_2d(x, weights, stride=1, padding='SAME', biases=None, relu=False, in_scale=None, in_shift=None,
            return_bn_stats=False):
    """tfw.conv_2d minus batch-norm (core.py:156-220).  x [B,H,W,Cin]; weights HWIO.
    With return_bn_stats the raw-output statistics for training-mode BN come back too."""
    x, weights = _f32(x, 'x'), _f32(weights, 'weights')
    B, H, W, Cin = x.shape
    kh, kw, cin2, cout = weights.shape
    assert cin2 == Cin
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    pad = {'VALID': 0, 'SAME': 1}[padding]
    if pad:
        Ho, Wo = -(-H // sh), -(-W // sw)
    else:
        Ho, Wo = (H - kh) // sh + 1, (W - kw) // sw + 1
    l = _lib.lib()
    y = torch.empty(B, Ho, Wo, cout, dtype=torch.float32, device=x.device)
    scratch = _scratch(l.sagen_conv2d_scratch_bytes(B, H, W, kh, kw, Cin, cout), x.device)
    stats = None
    if return_bn_stats:
        stats = torch.zeros(int(l.sagen_bn_stats_floats(B, Ho, Wo, cout)) // 2, dtype=torch.float64, device=x.device)
    check(l.sagen_conv2d(_ptr(x), B, H, W, Cin, _ptr(weights), kh, kw, cout, sh, sw, pad, _ptr(biases), int(relu),
                         _ptr(in_scale), _ptr(in_shift), _ptr(y), _ptr(stats), _ptr(scratch), scratch.numel() * 4, _stream()))
    return (y, stats) if return_bn_stats else y


def bn_finalize(stats, shape, gamma, beta, eps=1e-3):
    """contrib batch_norm, is_training=True (core.py:6,209-210): -> (scale, shift)."""
    B, Ho, Wo, C_ = shape
    scale = torch.empty(C_, dtype=torch.float32, device=stats.device)
    shift = torch.empty(C_, dtype=torch.float32, device=stats.device)
    check(_lib.lib().sagen_bn_finalize(_ptr(stats), B, Ho, Wo, C_, _ptr(_f32(gamma, 'gamma')), _ptr(_f32(beta, 'beta')),
                                       eps, _ptr(scale), _ptr(shift), _stream()))
    return scale, shift


def bn_apply_relu(x, scale=None, shift=None, residual=None):
    x = _f32(x, 'x')
    y = torch.empty_like(x)
    C_ = x.shape[-1]
    check(_lib.lib().sagen_bn_apply_relu(_ptr(x), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y), x.numel() // C_, C_, _stream()))
    return y


def maxpool3x3s2(x, scale=None, shift=None):
    x = _f32(x, 'x')
    B, H, W, C_ = x.shape
    y = torch.empty(B, (H + 1) // 2, (W + 1) // 2, C_, dtype=torch.float32, device=x.device)
    check(_lib.lib().sagen_maxpool3x3s2(_ptr(x), _ptr(scale), _ptr(shift), _ptr(y), B, H, W, C_, _stream()))
    return y


def fully_connected(x, weights, biases=None, relu=False):
    """tfw.fully_connected (core.py:43-93): acts on the last axis."""
    x, weights = _f32(x, 'x'), _f32(weights, 'weights')
    K, N = weights.shape
    M = x.numel() // K
    l = _lib.lib()
    y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    scratch = _scratch(l.sagen_fc_scratch_bytes(M, K, N), x.device)
    check(l.sagen_fc(_ptr(x), M, K, _ptr(weights), N, _ptr(biases), int(relu), _ptr(y), _ptr(scratch), scratch.numel() * 4, _stream()))
    return y


def deconv_2d(x, weights, stride, biases=None, relu=False):
    """tfw.deconv_2d (core.py:96-153), VALID.  weights [kh,kw,Cout,Cin]."""
    x, weights = _f32(x, 'x'), _f32(weights, 'weights')
    B, H, W, Cin = x.shape
    kh, kw, cout, cin2 = weights.shape
    assert cin2 == Cin
    sh, sw = stride
    l = _lib.lib()
    y = torch.empty(B, H * sh + kh - sh, W * sw + kw - sw, cout, dtype=torch.float32, device=x.device)
    scratch = _scratch(l.sagen_deconv2d_scratch_bytes(kh, kw, Cin, cout, sh, sw), x.device)
    check(l.sagen_deconv2d(_ptr(x), B, H, W, Cin, _ptr(weights), kh, kw, cout, sh, sw, _ptr(biases), int(relu), _ptr(y),
                           _ptr(scratch), scratch.numel() * 4, _stream()))
    return y


def mask_istft_mix(dmask, spec, coeffs):
    """sigmoid mask x STFT -> myutils.istft -> crop -> decoder sum (model.py:326-347, 421-434).
    dmask [B,28,1024,K]; spec [B,28,513,2]; coeffs [B,3,3,K+1] -> [B,4800,3]."""
    dmask, spec, coeffs = _f32(dmask, 'dmask'), _f32(spec, 'spec'), _f32(coeffs, 'coeffs')
    B, ntr = dmask.shape[0], dmask.shape[3]
    l = _lib.lib()
    out = torch.empty(B, 4800, 3, dtype=torch.float32, device=dmask.device)
    scratch = _scratch(l.sagen_mask_istft_mix_scratch_bytes(B), dmask.device)
    check(l.sagen_mask_istft_mix(_ptr(dmask), _ptr(spec), _ptr(coeffs), B, ntr, _ptr(out), _ptr(scratch), scratch.numel() * 4, _stream()))
    return out


def power_map(ambi_wyzx, sh_matrix):
    """AmbiDecoder.decode('projection') + per-direction RMS (decoder.py:24-28, distance.py:41-52)."""
    a, sh = _f32(ambi_wyzx, 'ambi'), _f32(sh_matrix, 'sh')
    T, P = a.shape[0], sh.shape[0]
    rms = torch.empty(P + 24 + (P % 2), dtype=torch.float32, device=a.device)
    check(_lib.lib().sagen_power_map(_ptr(a), T, _ptr(sh), P, _ptr(rms), _stream()))
    return rms[:P]


def power_map_batched(ambi_wyzx, sh_matrix):
    """One RMS map per chunk (SphericalAmbisonicsVisualizer.loop_frames, distance.py:41-59): ambi [n_chunks, T, 4] -> [n_chunks, P]."""
    a, sh = _f32(ambi_wyzx, 'ambi'), _f32(sh_matrix, 'sh')
    n, T, P = a.shape[0], a.shape[1], sh.shape[0]
    rms = torch.empty(n, P, dtype=torch.float32, device=a.device)
    moments = torch.empty(n * 10, dtype=torch.float64, device=a.device)
    check(_lib.lib().sagen_power_map_batched(_ptr(a), n, T, _ptr(sh), P, _ptr(rms), _ptr(moments), _stream()))
    return rms


def assemble_wyzx(audio, ambi_yzx, snd_contx=48000):
    """deploy.py:143-152: prepend W = mono[snd_contx/2 : snd_contx/2 + snd_dur]."""
    audio, ambi_yzx = _f32(audio, 'audio'), _f32(ambi_yzx, 'ambi')
    B, n = audio.shape[0], audio.shape[1]
    dur = ambi_yzx.shape[1]
    out = torch.empty(B, dur, 4, dtype=torch.float32, device=audio.device)
    check(_lib.lib().sagen_assemble_wyzx(_ptr(audio), _ptr(ambi_yzx), _ptr(out), B, n, snd_contx, dur, _stream()))
    return out


# ---- backward, op level (the gradients tf.gradients builds for the wrappers above; include/sagen.h) -----------------------
def wgrad(g, d, kh, kw, stride=(1, 1), origin=(0, 0), split=True):
    """dw[th,tw,cg,cd] = sum_{b,i,j} G[b, i*sh+th+h0, j*sw+tw+w0, :] (x) D[b,i,j,:].  conv_2d: G = x, D = dy, origin = -pad_before
    -> HWIO; deconv_2d: G = dy, D = x -> [kh,kw,Cout,Cin]; fully_connected: 2-D G [M,K], D [M,N] -> [K,N]."""
    g, d = _f32(g, 'g'), _f32(d, 'd')
    if g.dim() == 2:
        g, d = g[:, None, None, :], d[:, None, None, :]
    B, HG, WG, CG = g.shape
    _, HD, WD, CD = d.shape
    l = _lib.lib()
    dw = torch.empty(kh, kw, CG, CD, dtype=torch.float32, device=g.device)
    scratch = _scratch(l.sagen_wgrad_scratch_bytes(kh, kw, CG, CD), g.device) if split else None
    check(l.sagen_wgrad(_ptr(g), B, HG, WG, CG, _ptr(d), HD, WD, CD, kh, kw, stride[0], stride[1], origin[0], origin[1], _ptr(dw),
                        _ptr(scratch), scratch.numel() * 4 if split else 0, _stream()))
    return dw


def conv_2d_bwd_data(dy, weights, in_hw, stride=1, padding='SAME'):
    """Input gradient of conv_2d: dy [B,Ho,Wo,Cout], weights HWIO -> dx [B,H,W,Cin]."""
    dy, weights = _f32(dy, 'dy'), _f32(weights, 'weights')
    B, Ho, Wo, cout = dy.shape
    kh, kw, cin, cout2 = weights.shape
    assert cout2 == cout
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    H, W = in_hw
    l = _lib.lib()
    dx = torch.empty(B, H, W, cin, dtype=torch.float32, device=dy.device)
    scratch = _scratch(l.sagen_conv2d_bwd_data_scratch_bytes(kh, kw, cin, cout, sh, sw), dy.device)
    check(l.sagen_conv2d_bwd_data(_ptr(dy), B, Ho, Wo, cout, _ptr(weights), kh, kw, cin, sh, sw, {'VALID': 0, 'SAME': 1}[padding], H, W,
                                  _ptr(dx), _ptr(scratch), scratch.numel() * 4, _stream()))
    return dx


def bn_bwd(g, y, stats, gamma, beta, act=None, g2=None, eps=1e-3, want_dz=False):
    """Training-mode batch-norm backward at the raw conv output y [.., C] (stats from conv_2d(return_bn_stats=True)):
    dz = (g + g2) * (act > 0) -> (dy, dgamma, dbeta[, dz])."""
    g, y = _f32(g, 'g'), _f32(y, 'y')
    C_ = y.shape[-1]
    npix = y.numel() // C_
    dy = torch.empty_like(y)
    dz = torch.empty_like(y) if want_dz else None
    dgamma = torch.empty(C_, dtype=torch.float32, device=y.device)
    dbeta = torch.empty(C_, dtype=torch.float32, device=y.device)
    scratch = torch.empty((int(_lib.lib().sagen_bn_bwd_scratch_bytes(C_)) + 7) // 8, dtype=torch.float64, device=y.device)
    check(_lib.lib().sagen_bn_bwd(_ptr(g), _ptr(g2), _ptr(act), _ptr(y), _ptr(stats), _ptr(_f32(gamma, 'gamma')), _ptr(_f32(beta, 'beta')), eps,
                                  npix, C_, _ptr(dy), _ptr(dz), _ptr(dgamma), _ptr(dbeta), _ptr(scratch), scratch.numel() * 8, _stream()))
    return (dy, dgamma, dbeta, dz) if want_dz else (dy, dgamma, dbeta)


def maxpool3x3s2_bwd(y0, stats, gamma, beta, pooled, g, g2=None, eps=1e-3):
    """Backward of maxpool3x3s2(relu(bn(y0))) to the BN output: y0 [B,H,W,C] raw conv output, pooled / g [B,Ho,Wo,C]."""
    y0 = _f32(y0, 'y0')
    B, H, W, C_ = y0.shape
    dz = torch.empty_like(y0)
    check(_lib.lib().sagen_maxpool3x3s2_bwd(_ptr(y0), _ptr(stats), _ptr(_f32(gamma, 'gamma')), _ptr(_f32(beta, 'beta')), eps, _ptr(_f32(pooled, 'pooled')),
                                            _ptr(_f32(g, 'g')), _ptr(g2), _ptr(dz), B, H, W, C_, _stream()))
    return dz


def mask_istft_mix_bwd(dmask, spec, coeffs, dpred):
    """Adjoint of mask_istft_mix: dpred [B,4800,3] -> (d_dmask [B,28,1024,K], d_coeffs [B,3,3,K+1])."""
    dmask, spec, coeffs, dpred = _f32(dmask, 'dmask'), _f32(spec, 'spec'), _f32(coeffs, 'coeffs'), _f32(dpred, 'dpred')
    B, ntr = dmask.shape[0], dmask.shape[3]
    l = _lib.lib()
    dd = torch.empty_like(dmask)
    dc = torch.empty_like(coeffs)
    scratch = _scratch(l.sagen_mask_istft_mix_bwd_scratch_bytes(B, ntr), dmask.device)
    check(l.sagen_mask_istft_mix_bwd(_ptr(dmask), _ptr(spec), _ptr(coeffs), _ptr(dpred), B, ntr, _ptr(dd), _ptr(dc), _ptr(scratch),
                                     scratch.numel() * 4, _stream()))
    return dd, dc
