"""INTEGRATION.md section 2 - the ctypes stub a maintainer of the reference would paste next to deploy.py - executed VERBATIM: the code
block is cut out of the document and run in a fresh, torch-free Python process (device memory through hipMalloc of libamdhip64, the
null stream), for one deploy-sized audio-only batch, against the fp64 oracle."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import np_oracle as O
from spatialaudiogen_amd.weights import variable_specs, init_weights, synth_inputs
from util import rms, ensure_lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import sys, numpy as np
from collections import OrderedDict
exec(compile(open(sys.argv[1]).read(), 'sagen_binding.py', 'exec'))
assert 'torch' not in sys.modules
class P(object):
    encoders = ['audio']; separation = 'unet_mask'; num_sep_tracks = 32; loc_units = [512, 512]
    ambi_order = 1; audio_rate = 48000; video_rate = 10; context = 1.0; fft_window = 0.025
z = np.load(sys.argv[2])
ckpt = OrderedDict((k[4:], z[k]) for k in z.files if k.startswith('var/'))
m = HipW2XYZ(P(), ckpt, batch_size=10)
y = m.run(z['audio'])
y2 = m.run(z['audio'])
assert np.array_equal(y, y2)
np.save(sys.argv[3], y)
'''


def test_the_reference_side_ctypes_stub_runs_as_documented(tmp_path):
    import torch
    assert torch.cuda.is_available()
    ensure_lib()
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = doc[doc.index('## 2. The ctypes stub'):doc.index('## 3.')]
    blocks = re.findall(r'```python\n(.*?)```', sec, re.S)
    assert len(blocks) == 1 and 'class HipW2XYZ' in blocks[0] and "C.CDLL('libsagen_hip.so')" in blocks[0]
    stub = tmp_path / 'sagen_binding.py'
    stub.write_text(blocks[0])
    enc, B = ['audio'], 10
    Pv = init_weights(variable_specs(enc), seed=3, mode='test')
    inp = synth_inputs(B, enc, seed=5)
    np.savez(str(tmp_path / 'in.npz'), audio=inp['audio'], **{'var/' + k: v for k, v in Pv.items()})
    drv = tmp_path / 'drive.py'
    drv.write_text(DRIVER)
    env = dict(os.environ)
    # the stub loads both libraries by bare name: the loader path is the deployment's business (here: the in-tree library, ROCm's runtime)
    env['LD_LIBRARY_PATH'] = os.pathsep.join([os.path.join(ROOT, 'spatialaudiogen_amd'), '/opt/rocm/lib', env.get('LD_LIBRARY_PATH', '')])
    env.pop('PYTHONPATH', None)
    r = subprocess.run([sys.executable, str(drv), str(stub), str(tmp_path / 'in.npz'), str(tmp_path / 'out.npy')], env=env, cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(str(tmp_path / 'out.npy'))
    ref = O.SptAudioGenOracle(encoders=enc).inference_ops(inp['audio'], Pv)
    err = rms(got - ref)
    assert got.shape == (10, 4800, 3) and err <= 1e-4 and err <= 1e-3 * rms(ref), (err, rms(ref))
