"""Index geometry of the STFT -> encoder -> mask -> iSTFT path.

Every number here is derived with the same formulas the reference uses at graph-build
time; the cited lines are in /root/reference (model.py unless noted).
"""
from dataclasses import dataclass, field
import math
import numpy as np

from .definitions import AENC_FILTERS, AENC_KERNELS, AENC_STRIDES


def _conv_out(n, k, s):
    return (n - k) // s + 1          # VALID


def _deconv_out(n, k, s):
    return n * s + k - s             # core.py:139


@dataclass
class Geometry:
    audio_rate: int = 48000
    video_rate: int = 10
    context: float = 1.0
    sample_duration: float = 0.1
    ambi_order: int = 1
    fft_window: float = 0.025
    n_overlap: int = 4
    # derived
    snd_contx: int = field(init=False)
    snd_dur: int = field(init=False)
    snd_size: int = field(init=False)
    wind_size: int = field(init=False)
    hop: int = field(init=False)
    n_frames: int = field(init=False)
    enc_ss: int = field(init=False)
    enc_tt: int = field(init=False)
    mask_ss: int = field(init=False)
    mask_tt: int = field(init=False)
    dec_row0: int = field(init=False)
    dec_row1: int = field(init=False)
    istft_len: int = field(init=False)
    out_crop0: int = field(init=False)
    num_out: int = field(init=False)
    num_in: int = field(init=False)

    def __post_init__(self):
        assert float(self.audio_rate) / self.video_rate == int(self.audio_rate) // int(self.video_rate)  # model.py:33
        self.snd_contx = int(self.context * self.audio_rate)                     # model.py:39
        self.snd_dur = int(self.sample_duration * self.audio_rate)               # model.py:40
        self.snd_size = self.snd_contx + self.snd_dur - 1                        # model.py:41
        w = int(self.fft_window * self.audio_rate)                               # model.py:59
        self.wind_size = int(2 ** np.round(np.log2(w)))                          # model.py:60
        self.hop = self.wind_size // self.n_overlap
        n_winds = int(np.floor(self.snd_size / self.wind_size)) - 1              # myutils.py:126
        self.n_frames = n_winds * self.n_overlap
        inp_dim = 95.
        ss = (self.snd_contx / 2.) * (4. / self.wind_size)                       # model.py:166-171
        ss = int(ss - (inp_dim - 1) / 2.)
        tt = (self.snd_contx / 2. + self.snd_dur) * (4. / self.wind_size)
        tt = int(tt + (inp_dim - 1) / 2.)
        tt = int((np.ceil((tt - ss - inp_dim) / 16.)) * 16 + inp_dim + ss)
        self.enc_ss, self.enc_tt = ss, tt
        mss = np.floor((self.snd_contx / 2. - self.wind_size) * (4. / self.wind_size))       # model.py:314-318
        mtt = np.ceil((self.snd_contx / 2. + self.snd_dur + self.wind_size) * (4. / self.wind_size))
        skip = (self.snd_contx / 2.) * (4. / self.wind_size)
        skip = int(skip - (inp_dim - 1) / 2.)
        self.mask_ss, self.mask_tt = int(mss), int(mtt)
        self.dec_row0, self.dec_row1 = int(mss - skip), int(mtt - skip)          # model.py:323
        nfr = ((self.mask_tt - self.mask_ss) // self.n_overlap) * self.n_overlap  # myutils.py:187
        self.istft_len = (nfr // self.n_overlap) * self.wind_size - (self.n_overlap - 1) * self.hop
        ss2 = self.snd_contx / 2.                                                # model.py:344-347
        skip2 = np.floor((self.snd_contx / 2. - self.wind_size) * (4. / self.wind_size)) * (self.wind_size / 4.)
        skip2 += 3. * self.wind_size / 4.
        self.out_crop0 = int(ss2 - skip2)
        self.num_out = (self.ambi_order + 1) ** 2 - self.ambi_order ** 2         # model.py:242
        self.num_in = self.ambi_order ** 2                                       # model.py:243

    # shapes of the audio encoder pyramid (H = time frames, W = freq bins)
    def encoder_shapes(self):
        h, w = self.enc_tt - self.enc_ss, self.wind_size
        shapes = [(h, w, 1)]
        for nf, k, s in zip(AENC_FILTERS, AENC_KERNELS, AENC_STRIDES):
            h, w = _conv_out(h, k[0], s[0]), _conv_out(w, k[1], s[1])
            shapes.append((h, w, nf))
        return shapes

    def n_mask_frames(self):
        return self.mask_tt - self.mask_ss


DEFAULT = Geometry()
