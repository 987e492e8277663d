"""Constants of the path (same names/values as reference definitions.py:1-17)."""

AUDIO = 'audio'
VIDEO = 'video'
FLOW = 'flow'
ENCODERS = [AUDIO, VIDEO, FLOW]

NO_SEPARATION = 'none'
FREQ_MASK = 'unet_mask'
SEPARATION = [NO_SEPARATION, FREQ_MASK]

FFT_WINDOW = 25 * 0.001   # sec (eval metrics only)
FFT_OVERLAP_R = 2

NUM_SEP_TRACKS_DEF = 32
CTX_FEATS_FCUNITS_DEF = [64, 128, 128]
SEP_FREQ_MASK_FCUNITS_DEF = [256]
LOC_FCUNITS_DEF = [512, 512]
SEP_FFT_WINDOW_DEF = 0.025

# audio encoder / U-Net decoder architecture (reference model.py:162-164, 283-285)
AENC_FILTERS = [32, 64, 128, 256, 512]
AENC_KERNELS = [(7, 16), (3, 7), (3, 5), (3, 5), (3, 5)]
AENC_STRIDES = [(4, 8), (2, 4), (2, 2), (1, 1), (1, 1)]

# contrib batch_norm default epsilon (reference core.py:6,210 passes none)
BN_EPS = 1e-3
