"""Host-side logic: deploy window arithmetic (row 13), parameter-file parsing, sharding."""
import os
import numpy as np

from oracle import np_oracle as O
from spatialaudiogen_amd import deploy as D
from spatialaudiogen_amd.dist import shard_range


def test_window_table_reproduces_reference_float_quirks():
    """Transcription of feeder.py:64-72,121 + deploy.py:106-107: with audio_pow times
    0.5,0.6,... the shifted times are float64 values like 0.09999999999999998, so int() truncates: frame
    indices 0,0,1,3,4,5 (frame 2 never) and audio starts off by one in ~20% of the windows."""
    chunks = O.audio_pow_times(12)
    rows = O.deploy_window_table(chunks, 0., 10.)
    assert len(rows) == 95 and rows[-1][4] == 9                       # 95 windows, 10 groups (last has 5)
    assert [r[3] for r in rows[:8]] == [0, 0, 1, 3, 4, 5, 6, 7]
    assert [r[1] for r in rows[:8]] == [-24000, -19200, -14400, -9599, -4799, 0, 4800, 9599]
    assert [r[2] for r in rows[:6]] == [24000, 19200, 14400, 9599, 4799, 0]
    assert len(O.deploy_window_table(O.audio_pow_times(10), 0., 10.)) == 90


def test_deploy_helpers_match_the_oracle_table():
    chunks = O.audio_pow_times(12)
    ts = D.window_times(chunks, 0., 10.)
    rows = O.deploy_window_table(chunks, 0., 10.)
    assert ts == [r[0] for r in rows]
    assert [D.frame_index(t, 10) for t in ts] == [r[3] for r in rows]
    audio = (np.arange(12 * 48000, dtype=np.float64) + 1.0)[:, None]    # sample value = index + 1
    for t, start, pad_before, _, _, read_start in rows:
        w = D.audio_window(audio, t, 1.0, 52799, 48000)
        assert w.shape == (52799, 1)
        assert np.all(w[:pad_before] == 0)
        assert w[pad_before, 0] == read_start + 1                       # first real sample (feeder.py:81 quirk included)
        assert read_start in (max(start, 0), max(start, 0) - 1)
    late = D.audio_window(audio, 11.9, 1.0, 52799, 48000)              # runs past the end: zero post-padding
    assert late.shape == (52799, 1) and late[-1, 0] == 0 and late[0, 0] in (int(11.4 * 48000) + 1, int(11.4 * 48000))


def test_load_params_defaults_and_types(tmp_path):
    (tmp_path / 'train-params.txt').write_text(
        "encoders: ['audio', 'video']\nseparation: UNET_MASK\nambi_order: 1\naudio_rate: 48000\nvideo_rate: 10\n"
        "context: 1.0\nsample_dur: 0.1\nlr: 0.0001\n")
    p = D.load_params(str(tmp_path))
    assert p.encoders == ['audio', 'video'] and p.separation == 'unet_mask'
    assert p.num_sep_tracks == 64 and p.loc_units == [256, 256]       # legacy defaults (myutils.py:55-76)
    assert p.fft_window == 0.025 and p.context_units == [64, 128, 128] and p.freq_mask_units == []
    (tmp_path / 'train-params.txt').write_text(
        "encoders: ['audio']\nseparation: none\nambi_order: 1\naudio_rate: 48000\nvideo_rate: 10\ncontext: 1.0\n"
        "num_sep_tracks: 32\nloc_units: [512, 512]\nfreq_mask_units: []\n")
    p = D.load_params(str(tmp_path))
    assert p.num_sep_tracks == 32 and p.loc_units == [512, 512] and p.separation == 'none'


def test_shard_range_partitions_in_order():
    for n in (0, 1, 7, 8, 1024, 1023):
        for w in (1, 2, 3, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_ambisonics_host_module_matches_the_oracle_mesh():
    """Product-side mesh / order-1 harmonics (spatialaudiogen_amd/ambisonics.py) vs the oracle's restatement of
    distance.py:9-13 and common.py:151-157, and the analytic known answers of SURVEY.md 8c."""
    import numpy as np
    from spatialaudiogen_amd import ambisonics as A
    from oracle import np_oracle as O
    for res in (30.0, 20.0, 5.0):
        phi, nu = A.spherical_mesh(res)
        ophi, onu = O.spherical_mesh(res)
        assert np.array_equal(phi, ophi) and np.array_equal(nu, onu)
        assert np.allclose(A.sh_matrix(res), O.sh_matrix_order1(ophi, onu), atol=0, rtol=0)
    assert A.mesh_shape(30.0) == (7, 12) and A.mesh_shape(5.0) == (37, 72)
    assert np.allclose(A.sh_order1(np.pi / 2, 0.0), [1, 1, 0, 0]) and np.allclose(A.sh_order1(0.0, 0.0), [1, 0, 0, 1])
    assert np.allclose(A.sh_order1(0.3, np.pi / 2), [1, 0, 1, 0])


def test_eval_batches_do_not_depend_on_the_world_size():
    """evaluate.batch_shard: batches are cut from one global window order and whole batches are dealt to ranks, so the set of
    (batch -> windows) is identical for every world size; the trailing partial batch is dropped (or padded, last)."""
    from spatialaudiogen_amd.evaluate import batch_shard, BATCH_SIZE
    for n_windows in (0, 5, 16, 100, 1024 * 9, 1000):
        for mode in ('drop', 'pad'):
            ref = None
            for world in (1, 2, 3, 8):
                got = []
                for rank in range(world):
                    lo, hi, nb = batch_shard(n_windows, rank, world, mode)
                    got += list(range(lo, hi))
                assert got == list(range(nb))
                assert nb == n_windows // BATCH_SIZE + (1 if mode == 'pad' and n_windows % BATCH_SIZE else 0)
                ref = ref or got
                assert got == ref


def test_metric_reducer_keeps_the_reference_mean_and_reports_the_finite_one():
    """np.mean over all samples (eval.py:223): a NaN / inf sample shows in the mean, as in the reference; the finite-only
    statistics sit beside it."""
    import numpy as np
    from spatialaudiogen_amd.dist import MetricReducer
    red = MetricReducer(['a', 'b', 'c'])
    red.add_rows(np.array([[1.0, np.nan, 2.0], [3.0, 4.0, np.inf], [5.0, 6.0, 4.0]]))
    vals, n = red.reduce()
    assert n == 3 and vals['a'] == 3.0 and np.isnan(vals['b']) and np.isinf(vals['c'])
    assert red.finite_means == {'a': 3.0, 'b': 5.0, 'c': 3.0} and red.finite_counts == {'a': 3, 'b': 2, 'c': 2}


def test_bench_picks_a_group_size_that_divides_the_timed_steps():
    """bench.py --group 0 (auto): the largest divisor of K up to 10, so that the K timed steps are whole grouped calls; a K without a
    divisor of at least 4 takes min(K, 10) and a smaller call for the remainder; an explicit --group wins."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    # an even number of calls of 8 .. 16 batches where K allows it (two contexts in flight) ...
    assert b.pick_group(20, 0) == 10 and b.pick_group(30, 0) == 15 and b.pick_group(32, 0) == 16 and b.pick_group(40, 0) == 10
    assert b.pick_group(300, 0) == 15 and b.pick_group(24, 0) == 12 and b.pick_group(16, 0) == 8
    # ... else the largest divisor up to 10, and a remainder call where K has none worth having
    assert b.pick_group(50, 0) == 10 and b.pick_group(4, 0) == 4 and b.pick_group(2, 0) == 2 and b.pick_group(1, 0) == 1 and b.pick_group(3, 0) == 3
    assert b.pick_group(14, 0) == 7 and b.pick_group(17, 0) == 10 and b.pick_group(22, 0) == 11 and b.pick_group(23, 0) == 10      # 17 = 10 + 7, 22 = 2 x 11, 23 = 2 x 10 + 3
    assert b.pick_group(20, 3) == 3
    for k in range(1, 64):
        g = b.pick_group(k, 0)
        assert 1 <= g <= min(k, 16)
