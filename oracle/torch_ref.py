"""CPU ORACLE (second, independent implementation) — TEST INFRASTRUCTURE ONLY.

The same inference path as oracle/np_oracle.py, written against torch-CPU functional ops
(oneDNN) instead of numpy: used (a) to cross-check the numpy restatement op by op
(dual-implementation pin) and (b) as bench.py's `cpu_baseline` leg — the
"reference-equivalent CPU path" of BASELINE.md 3 (TF 1.4 cannot be installed here; this is
an upper bound on TF-1.4/Eigen CPU speed).  PARITY UNPINNED vs TF1 itself (see np_oracle.py).

Never imported by the product package.  Citations relative to /root/reference.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

AUDIO, VIDEO, FLOW = 'audio', 'video', 'flow'
BN_EPS = 1e-3


def _same(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2


def conv2d_tf(x, w_hwio, stride, padding):
    """x NCHW; w HWIO (core.py:184); TF SAME is asymmetric (extra pad after)."""
    w = w_hwio.permute(3, 2, 0, 1).contiguous()
    if padding == 'SAME':
        pt, pb = _same(x.shape[2], w.shape[2], stride[0])
        pl, pr = _same(x.shape[3], w.shape[3], stride[1])
        x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w, stride=stride)


def bn_train(x, gamma, beta):
    return F.batch_norm(x, None, None, gamma, beta, training=True, momentum=0.0, eps=BN_EPS)


def deconv2d_tf(x, w_hwoi, stride):
    """w [kh,kw,Cout,Cin] (core.py:118) -> torch conv_transpose2d weight [Cin,Cout,kh,kw]."""
    w = w_hwoi.permute(3, 2, 0, 1).contiguous()
    return F.conv_transpose2d(x, w, stride=stride)


class TorchRef(object):
    """Forward pass of SptAudioGen.inference_ops (model.py:356-434) on torch CPU tensors."""

    def __init__(self, P, encoders, sep_num_tracks=32, n_loc=2, dtype=torch.float32, device='cpu'):
        """device='cuda' runs the same restatement through torch-ROCm (test infrastructure for long trajectories: a few hundred
        fp64 autograd steps take minutes on the CPU); everything the product path is compared with stays THIS code."""
        self.dt = dtype
        self.dev = torch.device(device)
        self.P = {k: (v.detach().to(self.dev, dtype) if torch.is_tensor(v) else torch.as_tensor(np.asarray(v)).to(self.dev, dtype)) for k, v in P.items()}
        self.encoders = list(encoders)
        self.nsep = sep_num_tracks
        self.n_loc = n_loc
        self.W = 1024
        n = torch.arange(self.W, dtype=torch.float64)
        self.hann = (0.5 - 0.5 * torch.cos(2 * math.pi / self.W * n)).to(torch.float32).to(self.dev, dtype)
        self.ends = {}
        # optional {layer name: boolean NCHW / row mask}: ReLUs listed here are evaluated as x * mask instead of max(x, 0).  The
        # gradient tests pass the masks of the DEVICE activations: a ReLU whose input is within rounding distance of 0 may switch
        # differently in fp32 and fp64, and each such element changes its gradient by 100 % (a relative RMS error of
        # sqrt(fraction switched), 1e-3..1e-2 in the 1.6M-element ResNet layers) - a property of the comparison, not of the backward
        self.relu_masks = {}
        # record_masks=True: the switching pattern of THIS forward ({key: bool ndarray, same layout as relu_masks}) is kept in
        # own_masks - the tests bound how far the device's pattern is from it before they hand the device's pattern in
        self.record_masks = False
        self.own_masks = {}

    def _relu(self, x, key):
        if self.record_masks:
            self.own_masks[key] = (x.detach() > 0).cpu().numpy()
        m = self.relu_masks.get(key)
        return F.relu(x) if m is None else x * torch.as_tensor(m).to(x.device, x.dtype)

    def stft(self, audio):                     # [B, 52799] -> [B,200,1024] complex (myutils.py:119-147)
        fr = audio.unfold(1, self.W, self.W // 4)[:, :200]
        return torch.fft.fft(fr * self.hann, dim=-1)

    def resnet(self, x, scope):                # x [N,3,224,448] (resnet.py:123-236)
        P = self.P

        def cbn(x, name, s, act):
            y = conv2d_tf(x, P[name + '/weights'], (s, s), 'SAME')
            y = bn_train(y, P[name + '/bn/gamma'], P[name + '/bn/beta'])
            return self._relu(y, name) if act else y

        x = cbn(x, scope + '/conv1/conv', 2, True)
        pt, pb = _same(x.shape[2], 3, 2)
        pl, pr = _same(x.shape[3], 3, 2)
        x = F.max_pool2d(F.pad(x, (pl, pr, pt, pb), value=float('-inf')), 3, 2)
        for stage in (2, 3, 4, 5):
            for unit in (1, 2):
                name = '%s/conv%d_%d' % (scope, stage, unit)
                first = unit == 1 and stage > 2
                s = 2 if first else 1
                sc = conv2d_tf(x, P[name + '/shortcut/weights'], (2, 2), 'SAME') if first else x
                y = cbn(x, name + '/conv_1', s, True)
                y = cbn(y, name + '/conv_2', 1, False)
                x = self._relu(y + sc, name)
                self.ends[name] = x
        return x

    def fc(self, x, name, act=True):
        y = x @ self.P[name + '/weights'] + self.P[name + '/biases']
        return self._relu(y, name) if act else y

    @torch.no_grad()
    def forward(self, audio, video=None, flow=None):
        return self._forward(audio, video, flow)

    def _forward(self, audio, video=None, flow=None):
        P, dt = self.P, self.dt
        to_dev = lambda a: (a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))).to(self.dev, dt)
        audio = to_dev(audio)[:, :, 0]                                     # [B, 52799]
        B = audio.shape[0]
        S = self.stft(audio)                                               # [B,200,1024]
        mag = S[:, 46:173].abs().to(dt)[:, None]                           # NCHW [B,1,127,1024]
        ks = [(7, 16), (3, 7), (3, 5), (3, 5), (3, 5)]
        st = [(4, 8), (2, 4), (2, 2), (1, 1), (1, 1)]
        enc = [mag]
        x = mag
        for l in range(5):
            n = 'audio_encoder/conv%d' % (l + 1)
            x = F.relu(conv2d_tf(x, P[n + '/weights'], st[l], 'VALID') + P[n + '/biases'][None, :, None, None])
            enc.append(x)
            self.ends[n] = x
        # bottleneck (model.py:203-239): audio feature index w*512+c per time step
        a5 = enc[-1].permute(0, 2, 3, 1).reshape(B, 3, 6 * 512)
        feats = [self.fc(a5, 'bottleneck/audio-fc')]
        for k, inp in ((VIDEO, video), (FLOW, flow)):
            if k in self.encoders:
                v = to_dev(inp)[:, 0].permute(0, 3, 1, 2)
                v = self.resnet(v, k + '_encoder').permute(0, 2, 3, 1)      # NHWC [B,7,14,512]
                v = self.fc(v, 'bottleneck/%s-fc-red' % k).reshape(B, 1, 7 * 14 * 128)
                v = self.fc(v, 'bottleneck/%s-fc' % k)
                feats.append(v.expand(B, 3, 512))
        feats = torch.cat(feats, 2)                                        # [B,3,Cb]
        self.ends['bottleneck'] = feats
        # localization (model.py:241-271)
        x = feats
        for i in range(self.n_loc):
            x = self.fc(x, 'localization/fc%d' % (i + 1))
        x = self.fc(x, 'localization/fc%d' % (self.n_loc + 1), act=False).reshape(B, 3, 3, self.nsep + 1)
        self.ends['localization/coeffs'] = x
        w_loc, b_loc = x[..., :-1], x[..., -1]                             # [B,step,o,k], [B,step,o]
        # separation (model.py:282-348)
        f = self.fc(feats, 'separation/fc-feats')                          # [B,3,512]
        f = f.permute(0, 2, 1)[:, :, :, None].expand(B, 512, 3, 6)
        x = torch.cat([enc[-1], f], 1)
        for l in reversed(range(5)):
            n = 'separation/deconv%d' % (l + 1)
            x = deconv2d_tf(x, P[n + '/weights'], st[l]) + P[n + '/biases'][None, :, None, None]
            if l == 0:
                break
            x = torch.cat([F.relu(x), enc[l]], 1)
        self.ends['separation/deconv1'] = x
        m = torch.sigmoid(x[:, :, 43:71, :])                               # [B,32,28,1024]
        sep = S[:, None, 89:117, :] * m                                    # complex
        y = torch.fft.ifft(sep, dim=-1).real.to(dt)                        # [B,32,28,1024]
        ola = torch.zeros(B, self.nsep, 27 * 256 + 1024, dtype=dt, device=self.dev)
        for fidx in range(28):
            ola[:, :, fidx * 256:fidx * 256 + 1024] += y[:, :, fidx]
        xs = (ola / 4.)[:, :, 768:768 + 6400][:, :, 448:448 + 4800]        # myutils.py:196-205, model.py:344-347
        # decoder (model.py:421-434)
        step = torch.arange(4800, device=self.dev) // 1600
        w_t = w_loc[:, step]                                               # [B,4800,o,k]
        out = torch.einsum('bnok,bkn->bno', w_t, xs) + b_loc[:, step]
        return out


    # ---- training step checker (train.py:137-236): loss `stft/avg` (model.py:122-127, 156-159) and its gradient with
    #      respect to every trainable variable by autograd over the restated graph -------------------------------------
    def loss_and_grads(self, audio, video, flow, target, mask=None, keep=()):
        """fp64 recommended (dtype=torch.float64).  Returns (loss float, {variable name: gradient ndarray},
        prediction ndarray, {intermediate name: gradient ndarray for names in `keep`})."""
        names = [k for k in self.P if '/moving_' not in k]
        for k in names:
            self.P[k] = self.P[k].detach().clone().requires_grad_(True)
        with torch.enable_grad():
            pred = self._forward(audio, video, flow)
            kept = {}
            for k in keep:
                self.ends[k].retain_grad()
                kept[k] = self.ends[k]
            td = lambda a: (a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))).to(self.dev, self.dt)
            loss = stft_loss_torch(pred, td(target), None if mask is None else td(mask))
            loss.backward()
        grads = {k: self.P[k].grad.detach().cpu().numpy().copy() for k in names}
        igr = {k: v.grad.detach().cpu().numpy().copy() for k, v in kept.items()}
        for k in names:
            self.P[k] = self.P[k].detach()
        return float(loss.detach()), grads, pred.detach().cpu().numpy(), igr

    def loss_and_grad_tensors(self, audio, video, flow, target, mask=None):
        """As loss_and_grads but everything stays a tensor on self.dev: (loss tensor, {name: gradient tensor}) - the long-trajectory
        test keeps its whole Adam state on the device."""
        names = [k for k in self.P if '/moving_' not in k]
        for k in names:
            self.P[k] = self.P[k].detach().requires_grad_(True)
        with torch.enable_grad():
            pred = self._forward(audio, video, flow)
            td = lambda a: (a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))).to(self.dev, self.dt)
            loss = stft_loss_torch(pred, td(target), None if mask is None else td(mask))
            gs = torch.autograd.grad(loss, [self.P[k] for k in names])
        for k in names:
            self.P[k] = self.P[k].detach()
        return loss.detach(), dict(zip(names, gs))


def stft_for_loss_torch(x, window=2048, n_overlap=2):
    """myutils.stft_for_loss (myutils.py:151-178) for window = 2^ceil(log2(0.025*48000)) = 2048, 2 overlaps:
    x [B, N, C] -> complex [B, C, nW, window]."""
    B, N, C = x.shape
    n = torch.arange(window, dtype=torch.float64)
    hann = (0.5 - 0.5 * torch.cos(2 * math.pi / window * n)).to(torch.float32).to(x.device, x.dtype)
    wins = []
    stride = window // n_overlap
    for i in range(n_overlap):
        nW = int(float(N - i * stride - 1) / window)
        wins.append(x[:, i * stride:i * stride + window * nW, :].reshape(B, nW, window, C))
    w = torch.cat(wins, 1).permute(0, 3, 1, 2) * hann
    return torch.fft.fft(w, dim=-1)


def stft_loss_torch(pred, gt, mask=None):
    """losses['stft/mse'] = metrics['stft/avg'] (model.py:62-76, 122-127, 156-159)."""
    B, _, C = pred.shape
    if mask is None:
        mask = torch.ones(B, C, dtype=pred.dtype, device=pred.device)
    nm = torch.clamp(mask.sum(0), min=1.0)
    d = (stft_for_loss_torch(gt) - stft_for_loss_torch(pred)).abs() ** 2
    ps = d.mean(3).mean(2)                                   # [B, C]
    return ((ps * mask).sum(0) / nm * 100.).mean()
