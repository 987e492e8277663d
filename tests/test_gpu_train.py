"""Training-step pieces on the device (csrc/train.hip) against the oracle: the reference's loss `stft/avg` and its gradient
with respect to the prediction, and tf.train.AdamOptimizer over a flat bucket."""
import numpy as np
import pytest

from oracle import np_oracle as O
from util import rng, ensure_lib, rel_rms_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def T():
    import torch
    assert torch.cuda.is_available()
    ensure_lib()
    return torch


def test_stft_loss_value_and_gradient(T):
    from spatialaudiogen_amd.train import stft_loss
    r = rng(31)
    B = 4
    gt = 0.3 * r.normal(size=(B, 4800, 3))
    pred = gt * r.uniform(0.5, 1.2, size=(B, 1, 3)) + 0.05 * r.normal(size=gt.shape)
    mask = np.ones((B, 3)); mask[1, 1] = 0.0; mask[3, 1] = 0.0                 # WXY clips: no Z
    for mk in (None, mask):
        loss, grad = stft_loss(T.as_tensor(pred).cuda(), T.as_tensor(gt).cuda(), None if mk is None else T.as_tensor(mk))
        ref = O.stft_loss(pred, gt, mk)                                        # FFT form of model.py:62-76 (fp64)
        assert abs(float(loss) - ref) <= 1e-5 * abs(ref)
        probes = [(0, 0, 0), (0, 1023, 2), (1, 1024, 1), (1, 2047, 0), (2, 2048, 2), (2, 3071, 1), (3, 3072, 1), (3, 4095, 0), (3, 4096, 2),
                  (0, 4799, 1), (2, 700, 0), (1, 2500, 2)]
        fd = O.stft_loss_grad_fd(pred, gt, mk, probes)                         # finite differences of the FFT-form loss
        g = grad.cpu().numpy()
        got = np.array([g[p] for p in probes])
        assert np.abs(got - fd).max() <= 1e-4 * np.abs(fd).max() + 1e-9
        if mk is not None:
            assert np.all(g[1, :, 1] == 0) and np.all(g[3, :, 1] == 0)        # masked channels do not train
        assert np.all(g[:, 4096:, :] == 0)                                     # samples past the last full frame are outside every window


def test_adam_bucket_matches_tf_formulation(T):
    from spatialaudiogen_amd import train as tr
    specs = {'a/weights': (33, 7), 'b/biases': (5,), 'c/weights': (3, 3, 4, 8)}
    r = rng(5)
    P = {k: r.normal(size=s).astype(np.float32) for k, s in specs.items()}
    st = tr.AdamBuckets(specs, variables=P, lr=1e-2, lr_iters=3, lr_decay=0.5, bucket_bytes=1 << 10, device='cuda')
    ref = {k: (v.astype(np.float64), np.zeros(v.shape), np.zeros(v.shape)) for k, v in P.items()}
    for step in range(7):
        lr = O.exponential_decay_staircase(1e-2, step, 3, 0.5)
        for k in specs:
            g = r.normal(size=specs[k]).astype(np.float32) * (10.0 ** r.integers(-3, 2))
            st.view('grads', k).copy_(T.as_tensor(g))
            ref[k] = O.adam_tf(ref[k][0], g.astype(np.float64), ref[k][1], ref[k][2], step + 1, lr)
        assert st.apply() == pytest.approx(lr)
    for k in specs:
        assert rel_rms_err(st.view('params', k).cpu().numpy(), ref[k][0]) < 2e-6
        assert rel_rms_err(st.view('m', k).cpu().numpy(), ref[k][1]) < 2e-6 and rel_rms_err(st.view('v', k).cpu().numpy(), ref[k][2]) < 2e-6
