"""Step 2 of the TF1 pin kit - the ONE command that pins this repository's parity against the reference itself.

Runs on a machine that has what this image lacks: Python 2.7, tensorflow(-gpu)==1.4.0rc1 (the reference's requirements.txt:10) and a
checkout of pedro-morgado/spatialaudiogen.  It imports the REFERENCE's own `model.SptAudioGen`, builds `inference_ops`
(model.py:356-434) for the three cases of tools/make_golden.py, assigns the variables BY NAME from the .npz of step 1, runs the graph
on the stored inputs and writes the outputs in tests/golden/golden_v1.npz's layout:

    python2 tools/tf1_dump_golden.py /path/to/spatialaudiogen tf1_pin_inputs.npz tests/golden/tf1_v1.npz

Commit tests/golden/tf1_v1.npz (a few hundred KB of outputs - data, not source): tests/test_golden.py then holds BOTH the numpy oracle
and (on the GPU box) the HIP path to <= 1e-4 RMS of what TF1 computed, and the parity status moves from "unpinned" to "pinned".
Nothing of the reference is copied anywhere: it is imported from where it lies.  Written for Python 2.7 AND 3 syntax; numpy >= 1.14.

If the checkout lacks pyutils/tflib/models/image/resnet18.npy (the ImageNet initialisation, a binary the repository links to),
ResNet18.restore_pretrained - which only BUILDS initialisation assign ops that inference never runs (model.py:198-199; deploy.py
restores a checkpoint over them) - is skipped; every variable is assigned from the .npz either way."""
from __future__ import print_function
import os
import sys

import numpy as np

PROBE_KEYS = ['video_encoder/conv5_2', 'flow_encoder/conv5_2', 'separation/all_channels', 'decoder/ambix']


def checksum(v):
    """[sum, sum of squares, 16 strided probes] - the layout of tools/make_golden.py."""
    v = np.asarray(v, np.float64)
    flat = v.reshape(-1)
    return np.array([v.sum(), (v ** 2).sum()] + list(flat[:: max(v.size // 16, 1)][:16]))


def run_case(name, data, ref_dir):
    import tensorflow as tf
    import model as ref_model                      # the reference's model.py
    from pyutils.tflib.models.image import resnet as ref_resnet
    blob = os.path.join(os.path.dirname(os.path.abspath(ref_resnet.__file__)), 'resnet18.npy')
    if not os.path.exists(blob):
        print('note: %s is missing - skipping the construction of the ImageNet initialisation ops (never run by inference)' % blob)
        ref_resnet.ResNet18.restore_pretrained = lambda self, *a, **k: []
    enc = str(data[name + '/encoders']).split(',')
    tf.reset_default_graph()
    feeds, ph = {}, {}
    for key in ('audio', 'video', 'flow'):
        k = '%s/in/%s' % (name, key)
        if k in data:
            x = np.asarray(data[k], np.float32)
            ph[key] = tf.placeholder(dtype=tf.float32, shape=x.shape)
            feeds[ph[key]] = x
    params = ref_model.SptAudioGenParams(sep_num_tracks=32, ctx_feats_fc_units=[64, 128, 128], loc_fc_units=[512, 512],
                                         sep_freq_mask_fc_units=[], sep_fft_window=0.025)
    net = ref_model.SptAudioGen(ambi_order=1, audio_rate=48000, video_rate=10, context=1., sample_duration=0.1,
                                encoders=enc, separation='unet_mask', params=params)
    x_ambi = net.inference_ops(is_training=False, **ph)                    # as deploy.py:75 calls it
    out = {}
    with tf.Session(config=tf.ConfigProto(allow_soft_placement=True)) as sess:
        sess.run(tf.global_variables_initializer())
        assigned, left = 0, []
        for var in tf.global_variables():
            k = '%s/var/%s' % (name, var.op.name)
            if k in data:
                val = np.asarray(data[k], np.float32)
                assert tuple(val.shape) == tuple(var.get_shape().as_list()), (var.op.name, val.shape, var.get_shape().as_list())
                var.load(val, sess)
                assigned += 1
            else:
                left.append(var.op.name)
        # what may stay at its initial value: batch-norm moving averages when the .npz carries none (never read: BN runs on batch statistics)
        bad = [n for n in left if '/moving_' not in n]
        assert not bad, 'graph variables without a value in the .npz: %s' % bad[:8]
        provided = [k for k in data.keys() if k.startswith(name + '/var/')]
        assert assigned == len(provided), 'the .npz holds %d variables of which the graph took %d' % (len(provided), assigned)
        fetch = {'ambix': x_ambi}
        for key in PROBE_KEYS:
            if key in net.ends:
                fetch['chk/' + key] = net.ends[key]
        got = sess.run(fetch, feed_dict=feeds)
    out[name + '/ambix'] = np.asarray(got['ambix'], np.float32)
    if name + '/fingerprint' in data:                                      # ties the outputs to the weights / inputs they came from
        out[name + '/fingerprint'] = np.asarray(data[name + '/fingerprint'], np.float64)
    for k, v in got.items():
        if k.startswith('chk/'):
            out['%s/%s' % (name, k)] = checksum(v)
            out['%s/shape/%s' % (name, k[4:])] = np.array(np.asarray(v).shape)
    print('case %s: ambix %s rms %.6g' % (name, out[name + '/ambix'].shape, float(np.sqrt(np.mean(out[name + '/ambix'].astype(np.float64) ** 2)))))
    return out


def main(argv):
    if len(argv) != 4:
        print(__doc__)
        return 2
    ref_dir, inp_fn, out_fn = argv[1:]
    sys.path.insert(0, os.path.abspath(ref_dir))
    z = np.load(inp_fn)
    data = dict((k, z[k]) for k in z.files)
    cases = sorted(set(k.split('/')[0] for k in data if k.endswith('/encoders')))
    out = {}
    for name in cases:
        out.update(run_case(name, data, ref_dir))
    import tensorflow as tf
    out['tf_version'] = np.array(tf.__version__)
    np.savez_compressed(out_fn, **out)
    print('wrote %s (%d arrays)' % (out_fn, len(out)))
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv))
