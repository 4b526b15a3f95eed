"""GPU parity: Iterative-F0 (reference method 3) through the C ABI vs fixtures made by the
reference's own code and vs the oracle.  fp64.  The summary spectra are held to 1e-9; the chroma to the
north_star 1e-5 (the saliences are ~1e13 because of quirk A.11, and the period search is discrete)."""
import json
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FS = 22050
SPELLINGS = (("unicode", ""), ("ascii", "_ascii"))   # note_names mode, fixture key suffix (make_golden.py)


@pytest.fixture(scope="module")
def eng():
    import chord_detection_amd as cd
    return cd.get_engine(0)


@pytest.fixture(scope="module")
def clips(golden_dir):
    d = np.load(os.path.join(golden_dir, "clips.npz"))
    return {k: d[k] for k in d.files if k != "fs"}


def test_golden_clips(eng, clips, golden_dir):
    import chord_detection_amd as cd
    d = np.load(os.path.join(golden_dir, "iterative_f0.npz"))
    assert cd.METHODS[3] is cd.MultipitchIterativeF0 and list(cd.METHODS.keys()) == [1, 2, 3, 4]
    assert cd.MultipitchIterativeF0.display_name() == "Iterative F0 (Klapuri, Anssi)"
    expected = json.load(open(os.path.join(golden_dir, "constants.json")))["test_py_expected"]
    for name in ("tone_Csharp3", "tone_E4", "tones_G3_Asharp4", "tones_G2_B2_Gsharp3", "piano_like_Cmaj", "poly_seed1",
                 "poly_seed2", "short_ragged"):
        x = clips[name]
        ut = eng.iterative_f0_spectra(x, FS)
        np.testing.assert_allclose(ut[0][:512], d[name + "/ut0_head"], rtol=1e-9)
        for mode, sfx in SPELLINGS:
            total, per = eng.iterative_f0(x, FS, return_frames=True, note_names=mode)
            np.testing.assert_allclose(per, d[name + "/frames" + sfx], rtol=1e-5, atol=0)
            np.testing.assert_allclose(total, d[name + "/sum" + sfx], rtol=1e-5, atol=0)
            c = cd.MultipitchIterativeF0((x, FS), note_names=mode).compute_pitches()
            assert repr(c) == str(d[name + "/repr" + sfx])
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                assert c.key() == str(d[name + "/key" + sfx])
            if mode == "unicode":
                assert all(total[pc] == 0.0 for pc in (1, 3, 6, 8, 10))   # quirk A.18
            if name in expected:
                print("if0    %-8s %-20s engine %s  tests/test.py expects %s" % (mode, name, repr(c), expected[name]))
    got = eng.iterative_f0_batch([clips["tone_Csharp3"], clips["tone_E4"]], FS, note_names="ascii")
    np.testing.assert_allclose(got[0], d["tone_Csharp3/sum_ascii"], rtol=1e-5)
    np.testing.assert_allclose(got[1], d["tone_E4/sum_ascii"], rtol=1e-5)
    assert got[0][1] > 0      # the C# bin of the C#3 tone survives with ASCII names
    with pytest.raises(ValueError):
        eng.iterative_f0(clips["tone_E4"], FS, note_names="latin1")


def test_spectra_and_batch_vs_oracle(eng, clips):
    from oracle import iterative_f0 as o_if0
    x = clips["poly_seed1"][:20000]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        per, Ut = o_if0.iterative_f0_frames(x, FS)
    ut = eng.iterative_f0_spectra(x, FS)
    assert ut.shape == Ut.shape == (3, 16384)
    np.testing.assert_allclose(ut, Ut, rtol=1e-9, atol=1e-9 * np.abs(Ut).max())
    batch = [x, np.zeros(0, dtype=np.float32), clips["short_ragged"], np.zeros(9000, dtype=np.float32)]
    got = eng.iterative_f0_batch(batch, FS)
    np.testing.assert_allclose(got[0], per.sum(0), rtol=1e-5)
    assert np.all(got[1] == 0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_allclose(got[2], o_if0.iterative_f0_compute(clips["short_ragged"], FS), rtol=1e-5)
        np.testing.assert_allclose(got[3], o_if0.iterative_f0_compute(batch[3], FS), rtol=1e-5, atol=0)
    assert np.array_equal(got, eng.iterative_f0_batch(batch, FS))      # deterministic
    with pytest.raises(NotImplementedError):
        eng.iterative_f0(x, FS, frame_size=16385)      # above 16384 samples there is no kernel (round 6: 8193 ... 16384 run)
    with pytest.raises(NotImplementedError):
        eng.iterative_f0(x, FS, frame_size=8)
    with pytest.raises(ValueError):
        eng.iterative_f0(np.zeros((2, 2), dtype=np.float32), FS)


def test_periodicity_estimator_attribute(eng, clips):
    """iterative_f0.py:44: MultipitchIterativeF0.periodicity_estimator is an IterativeF0PeriodicityAnalysis whose
    compute(Uk) -> (Chromagram, (voicesaliences, voiceperiods)) runs the period search on one summary spectrum (periodicity.py:48-163):
    here on the oracle's summary spectra, frame by frame, against the oracle's period search, both spellings; and on the
    engine's own spectra against the engine's per-frame rows."""
    import chord_detection_amd as cd
    from oracle import iterative_f0 as o_if0
    x = clips["poly_seed1"]
    for mode, _ in SPELLINGS:
        obj = cd.MultipitchIterativeF0((x, FS), note_names=mode)
        est = obj.periodicity_estimator
        assert est.fs == FS and est.window_size == 8192 and est.max_voices == 4 and est.Q == 20
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wper, wut = o_if0.iterative_f0_frames(x, FS, note_names=mode)
        o_est = o_if0.Periodicity(FS, 8192, note_names=mode)   # one object for the clip, like iterative_f0_frames (smax leaks)
        for f in range(wut.shape[0]):
            c, plots = est.compute(wut[f])
            assert len(c) == 12
            np.testing.assert_allclose(c.as_array(), wper[f], rtol=1e-5, atol=1e-300)
            # periodicity.py:112: (voicesaliences.copy(), voiceperiods.copy()) -- max_voices entries each, zeros for the voices
            # the search did not find; also attributes of the estimator, like the reference's
            sal, per = plots
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                _, wsal, wtau = o_est.compute(wut[f])
            assert sal.shape == per.shape == (4,)
            np.testing.assert_allclose(sal, wsal, rtol=1e-5, atol=1e-300)
            np.testing.assert_allclose(per, wtau, rtol=1e-9, atol=0)
            assert np.array_equal(est.voicesaliences, sal) and np.array_equal(est.voiceperiods, per)
            assert np.array_equal(per == 0, wtau == 0) and (per[0] > 0)
        with pytest.raises(ValueError):
            est.compute(wut[:2])                      # periodicity.py:48 takes ONE spectrum
        ut = eng.iterative_f0_spectra(x, FS)
        _, per = eng.iterative_f0(x, FS, return_frames=True, note_names=mode)
        np.testing.assert_array_equal(eng.iterative_f0_periodicity(ut, FS, note_names=mode), per)
    with pytest.raises(ValueError):
        eng.iterative_f0_periodicity(np.ones((1, 1000)), FS, frame_size=8192)


@pytest.mark.parametrize("fs,frame_size,f0s", [(44100, 1024, (41.2, 43.0)), (22050, 512, (55.0, 82.4)), (22050, 300, (65.4, 98.0)),
                                                (44100, 8192, (41.2, 46.2))])
def test_harmonic_cancellation_with_overlapping_partial_windows(eng, fs, frame_size, f0s):
    """periodicity.py:83-96 adds the 9-bin windows of the partials one after the other.  When the partials are less than
    nine bins apart (K / tau < 9: f0 below ~48 Hz at the default frame, ordinary bass notes at small frames -- below ONE
    bin at 300 / 512 samples) the windows overlap; the kernel then gathers per bin in ascending m (round-4 advisor
    finding: the scatter it replaced could lose updates).  Low tones, against the oracle's sequential loop."""
    from oracle import iterative_f0 as o_if0
    rng = np.random.default_rng(int(fs + frame_size))
    n = 6 * frame_size + 17
    t = np.arange(n) / fs
    x = np.zeros(n)
    for f0 in f0s:
        for h in range(1, 9):
            x += 0.7 ** h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6))
    x = (0.15 * x + 1e-3 * rng.standard_normal(n)).astype(np.float32)
    kw = dict(frame_size=frame_size)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wper, wut = o_if0.iterative_f0_frames(x, fs, note_names="ascii", **kw)
    tot, per = eng.iterative_f0(x, fs, return_frames=True, note_names="ascii", **kw)
    assert per.shape == wper.shape and np.abs(wper).sum() > 0
    np.testing.assert_allclose(per, wper, rtol=1e-5, atol=1e-300)
    # the estimator alone on the oracle's spectra: the cancellation is all that differs between voices
    np.testing.assert_allclose(eng.iterative_f0_periodicity(wut, fs, note_names="ascii", **kw), wper, rtol=1e-5, atol=1e-300)


def test_chunked_front_end_matches_sequential_filtering(eng):
    """A signal longer than one 262144-sample chunk: chunks start from zero state mpx_iterative_f0_warmup (40960) samples early.
    The reference filters sequentially; the oracle does too.  Agreement shows the run-in is long enough."""
    from oracle import iterative_f0 as o_if0
    rng = np.random.default_rng(11)
    n = 262144 + 3 * 8192 + 1000
    t = np.arange(n) / FS
    x = (0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 331 * t) + 0.02 * rng.standard_normal(n)).astype(np.float32)
    ut = eng.iterative_f0_spectra(x, FS)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Ut = o_if0.summary_spectra(x, FS)
    assert ut.shape == Ut.shape
    # frames 32.. live in the second chunk
    np.testing.assert_allclose(ut[30:], Ut[30:], rtol=1e-9, atol=1e-9 * np.abs(Ut).max())


def _poly(rng, n, fs=FS, voices=3):
    t = np.arange(n) / fs
    x = np.zeros(n)
    for _ in range(voices):
        f0 = 440.0 * 2.0 ** ((int(rng.integers(40, 80)) - 69) / 12.0)
        for h in range(1, 5):
            x += 0.6 ** h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6))
    return (0.2 * x).astype(np.float32)


def test_clip_list_above_the_cap_runs_in_time_slices_with_every_clip_in_flight():
    """A clip list whose front-end output exceeds the cap while ONE frame of every clip fits is not halved: it runs in time
    slices with every clip in flight (the corpus driver hands Iterative-F0 a whole 4096-clip group this way).  300 ragged
    clips of ~20 000 samples under a 1 GiB cap (3.4 GB of hand-off: three frames per clip, so one or two frames per slice):
    rows equal to the uncapped call bit for bit, and to the oracle."""
    import chord_detection_amd as cd
    from oracle import iterative_f0 as o_if0
    rng = np.random.default_rng(31)
    base = [_poly(rng, 20000) for _ in range(7)]
    batch = [base[i % 7][: 20000 - 11 * (i % 13)] for i in range(300)]
    whole, cut = cd.Engine(0), cd.Engine(0)
    try:
        cut.set_option("if0_workspace_bytes", 1 << 30)
        assert sum(-(-len(b) // 8192) * 8192 for b in batch) * 70 * 8 > 3 * (1 << 30)
        got_cut = cut.iterative_f0_batch(batch, FS)
        got_whole = whole.iterative_f0_batch(batch, FS)
        np.testing.assert_array_equal(got_cut, got_whole)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for i in (0, 149, 299):
                np.testing.assert_allclose(got_cut[i], o_if0.iterative_f0_compute(batch[i], FS), rtol=1e-5, atol=0)
    finally:
        cut.close()
        whole.close()


def test_large_batch_is_split_without_changing_results():
    """More clips than fit one front-end workspace (8 B x 70 channels per sample; the cap is 32 GiB by default, 64 MiB
    here through mpx_set_option(MPX_OPT_IF0_WORKSPACE_BYTES) -- the release library's own route, no environment knob).  One
    frame of each of the 23 clips is 105 MB: the list is halved (one frame per clip must fit for time slices), and each half
    -- 12 / 11 clips, 55 / 50 MB per frame, 165 MB in all -- then runs in time slices.  Every clip must come out exactly as
    from the uncapped call, and as the oracle computes it."""
    import chord_detection_amd as cd
    from oracle import iterative_f0 as o_if0
    rng = np.random.default_rng(3)
    base = [_poly(rng, 20000) for _ in range(6)]
    batch = [base[i % 6][: 20000 - 7 * (i % 5)] for i in range(23)]          # ragged lengths: the split points move
    whole = cd.Engine(0)
    cut = cd.Engine(0)
    try:
        assert cut.get_option("if0_workspace_bytes") == 32 << 30
        cut.set_option("if0_workspace_bytes", 64 << 20)
        assert cut.get_option("if0_workspace_bytes") == 64 << 20
        with pytest.raises(ValueError):
            cut.set_option("if0_workspace_bytes", 1 << 20)
        need = sum(-(-len(b) // 8192) * 8192 for b in batch) * 70 * 8
        assert need > 4 * (64 << 20)
        got_cut = cut.iterative_f0_batch(batch, FS)
        got_whole = whole.iterative_f0_batch(batch, FS)
        np.testing.assert_array_equal(got_cut, got_whole)                     # split == unsplit, bit for bit
        packed = np.stack([b[:19972] for b in batch])                         # [23, 19972]: the packed fast path as well
        np.testing.assert_array_equal(cut.iterative_f0_batch(packed, FS), whole.iterative_f0_batch(packed, FS))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for i in (0, 11, 12, 22):                                         # either side of the first split point, and the ends
                np.testing.assert_allclose(got_cut[i], o_if0.iterative_f0_compute(batch[i], FS), rtol=1e-5, atol=0)
    finally:
        cut.close()
        whole.close()


@pytest.mark.parametrize("frame_size,channels", [(8192, 70), (1024, 70), (2048, 33), (8192, 64)])
def test_time_slices_carry_the_filter_state_exactly(frame_size, channels):
    """One clip whose front-end output exceeds the workspace cap (64 MiB here, the smallest the ABI takes): every chunk
    advances in time slices of whole frames, the filter state of every lane travels from launch to launch (If0Slice), the
    hand-off buffer holds one slice.  Same operations on the same operands: summary spectra and chroma equal the one-piece
    call BIT FOR BIT -- with 64 + 6 channels (a full wave and a wave of leftover channels per chunk), 33 (leftovers only) and
    64 (none), frames of 8192 / 2048 / 1024 -- and the oracle's sequential filtering to 1e-9 / 1e-5."""
    import chord_detection_amd as cd
    from oracle import iterative_f0 as o_if0
    rng = np.random.default_rng(77 + frame_size + channels)
    n = 300000 + 1234
    x = _poly(rng, n) + (1e-3 * rng.standard_normal(n)).astype(np.float32)
    kw = dict(frame_size=frame_size, channels=channels)
    whole, cut = cd.Engine(0), cd.Engine(0)
    try:
        cut.set_option("if0_workspace_bytes", 64 << 20)
        assert n * channels * 8 > 64 << 20                      # more than one slice
        ut_w = whole.iterative_f0_spectra(x, FS, **kw)
        ut_c = cut.iterative_f0_spectra(x, FS, **kw)
        np.testing.assert_array_equal(ut_c, ut_w)
        tot_w, per_w = whole.iterative_f0(x, FS, return_frames=True, **kw)
        tot_c, per_c = cut.iterative_f0(x, FS, return_frames=True, **kw)
        np.testing.assert_array_equal(per_c, per_w)
        np.testing.assert_array_equal(tot_c, tot_w)
        # the batch entry point with one oversized clip among small ones: the list is halved down to the clip, which is sliced
        batch = [x[:9000], x, x[:20000]]
        np.testing.assert_array_equal(cut.iterative_f0_batch(batch, FS, **kw), whole.iterative_f0_batch(batch, FS, **kw))
    finally:
        cut.close()
        whole.close()
    if frame_size == 8192 and channels == 70:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wper, wut = o_if0.iterative_f0_frames(x[:60000], FS, **kw)
        nfr = wper.shape[0] - 1                               # the oracle's last frame is the zero-padded tail of the cut
        np.testing.assert_allclose(ut_c[:nfr], wut[:nfr], rtol=1e-9, atol=1e-9 * np.abs(wut).max())
        np.testing.assert_allclose(per_c[:nfr], wper[:nfr], rtol=1e-5, atol=1e-300)


@pytest.mark.parametrize("frame_size,power,channels", [(1000, 1.0, 70), (1500, 0.5, 70), (2047, 1.0, 33), (3000, 1.0, 70),
                                                       (4095, 2.0, 64), (512, 1.0, 70), (777, 1.0, 70),
                                                       (4097, 1.0, 70), (5000, 0.5, 70), (6000, 1.0, 33), (8191, 1.0, 70), (8190, 2.0, 64)])
def test_any_frame_size_by_chirp_z_vs_oracle(eng, frame_size, power, channels):
    """iterative_f0.py:25 takes any integer frame_size.  Sizes other than 1024 / 2048 / 4096 / 8192 -- up to 8191 samples --
    run the 2 x frame_size-point spectrum as a chirp-z transform (if0_spectrum_blue_kernel: 4096 points up to 2048 samples,
    8192 up to 4095; if0_spectrum_blue2_kernel: 16384 points as two residues of 8192 above) on front-end chunks of
    lcm(frame_size, 64) x k samples: odd, even and prime-ish sizes, a power of two below 1024, all three transform lengths,
    a clip of several chunks, against the oracle (spectra 1e-9, per-frame chroma 1e-5)."""
    from oracle import iterative_f0 as o_if0
    rng = np.random.default_rng(5000 + frame_size)
    kw = dict(frame_size=frame_size, power=power, channels=channels)
    for n in (2 * frame_size + frame_size // 3 + 5, 150000 if frame_size in (1000, 3000, 6000) else 0):
        if not n:
            continue
        x = _poly(rng, n) + (1e-3 * rng.standard_normal(n)).astype(np.float32)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wper, wut = o_if0.iterative_f0_frames(x, FS, **kw)
        ut = eng.iterative_f0_spectra(x, FS, **kw)
        tot, per = eng.iterative_f0(x, FS, return_frames=True, **kw)
        assert ut.shape == wut.shape == (-(-n // frame_size), 2 * frame_size)
        np.testing.assert_allclose(ut, wut, rtol=1e-9, atol=1e-9 * np.abs(wut).max())
        np.testing.assert_allclose(per, wper, rtol=1e-5, atol=1e-300)
        np.testing.assert_allclose(tot, wper.sum(0), rtol=1e-5, atol=1e-300)
    batch = [x[:9000], np.zeros(0, dtype=np.float32), x[:frame_size - 1]]
    got = eng.iterative_f0_batch(batch, FS, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_allclose(got[0], o_if0.iterative_f0_compute(batch[0], FS, **kw), rtol=1e-5, atol=1e-300)
        np.testing.assert_allclose(got[2], o_if0.iterative_f0_compute(batch[2], FS, **kw), rtol=1e-5, atol=1e-300)
    assert np.all(got[1] == 0)


@pytest.mark.parametrize("frame_size,power,channels,fs", [(12000, 1.0, 70, 22050), (16384, 1.0, 70, 22050), (8193, 0.5, 33, 22050),
                                                          (16383, 1.0, 64, 44100), (10000, 2.0, 70, 44100)])
def test_frame_sizes_above_8192_vs_oracle(eng, frame_size, power, channels, fs):
    """Round 6: iterative_f0.py:25 takes any integer frame_size; 8193 ... 16384 samples run the 2 x frame_size-point spectrum
    as a chirp-z transform of 32768 points -- four residues of 8192 around a radix-4 step (if0_spectrum_blue4_kernel) -- and the
    period search on spectra of up to 32768 bins (if0_periodicity_kernel<true>: twice the tables).  The review's two sizes
    (12000, 16384), the ends of the range, both sample rates, three powers; spectra 1e-9, per-frame chroma 1e-5; the
    estimator object of the drop-in class at such a window; a batch."""
    import chord_detection_amd as cd
    from oracle import iterative_f0 as o_if0
    rng = np.random.default_rng(9000 + frame_size)
    kw = dict(frame_size=frame_size, power=power, channels=channels)
    n = 2 * frame_size + frame_size // 3 + 5
    x = _poly(rng, n) + (1e-3 * rng.standard_normal(n)).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wper, wut = o_if0.iterative_f0_frames(x, fs, **kw)
    ut = eng.iterative_f0_spectra(x, fs, **kw)
    tot, per = eng.iterative_f0(x, fs, return_frames=True, **kw)
    assert ut.shape == wut.shape == (3, 2 * frame_size)
    np.testing.assert_allclose(ut, wut, rtol=1e-9, atol=1e-9 * np.abs(wut).max())
    np.testing.assert_allclose(per, wper, rtol=1e-5, atol=1e-300)
    np.testing.assert_allclose(tot, wper.sum(0), rtol=1e-5, atol=1e-300)
    assert np.array_equal(eng.iterative_f0_periodicity(ut, fs, frame_size=frame_size), per)     # the search alone, same bits
    batch = [x[:frame_size + 9], np.zeros(0, dtype=np.float32), x[:frame_size - 1]]
    got = eng.iterative_f0_batch(batch, fs, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_allclose(got[0], o_if0.iterative_f0_compute(batch[0], fs, **kw), rtol=1e-5, atol=1e-300)
        np.testing.assert_allclose(got[2], o_if0.iterative_f0_compute(batch[2], fs, **kw), rtol=1e-5, atol=1e-300)
    assert np.all(got[1] == 0)
    if frame_size == 12000:
        obj = cd.MultipitchIterativeF0((x, fs), frame_size=frame_size)
        np.testing.assert_allclose(obj.compute_pitches().as_array(), wper.sum(0) if channels == 70 and power == 1.0 else tot, rtol=1e-5, atol=1e-300)
        c, (sal, tau) = obj.periodicity_estimator.compute(wut[0])
        np.testing.assert_allclose(c.as_array(), wper[0], rtol=1e-5, atol=1e-300)
        assert sal.shape == (4,) and tau[0] > 0


@pytest.mark.parametrize("frame_size,nfr,channels", [(900, 49, 70), (1000, 13, 70), (777, 30, 64), (1500, 9, 33)])
def test_chirp_z_sequential_front_end_flushes_the_last_partial_tile(eng, frame_size, nfr, channels):
    """Packed clips whose length is an exact multiple of a chirp-z frame size that is not a multiple of 16: the clip's last
    chunk ends inside a 16-sample tile.  With more waves than SIMDs (640 clips) the front end is the SEQUENTIAL kernel
    (if0_frontend2_kernel), which once dropped that tile (round-4 advisor finding): every row against the single-clip call
    (pipelined kernel), and against the oracle for the distinct clips (iterative_f0.py:54-96)."""
    from oracle import iterative_f0 as o_if0
    rng = np.random.default_rng(frame_size)
    n = frame_size * nfr
    assert n % 16 != 0
    base = [_poly(rng, n) + (1e-3 * rng.standard_normal(n)).astype(np.float32) for _ in range(4)]
    kw = dict(frame_size=frame_size, channels=channels)
    packed = np.stack([base[i % 4] for i in range(640)])
    got = eng.iterative_f0_batch(packed, FS, **kw)
    for i in range(4):
        one = eng.iterative_f0(base[i], FS, **kw)
        np.testing.assert_allclose(got[i::4], np.broadcast_to(one, got[i::4].shape), rtol=1e-12, atol=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in (0, 3):
            np.testing.assert_allclose(got[636 + i], o_if0.iterative_f0_compute(base[i], FS, **kw), rtol=1e-5, atol=1e-300)


@pytest.mark.parametrize("frame_size", [1024, 2048, 4096])
@pytest.mark.parametrize("power", [0.5, 1.0, 2.0])
@pytest.mark.parametrize("channels", [31, 70])
def test_frame_sizes_powers_channels_vs_oracle(eng, frame_size, power, channels):
    """iterative_f0.py:22-33 takes frame_size, power and channels; every instantiation of the summary-spectrum kernel a
    caller can reach (frame 1024 / 2048 / 4096 next to the default 8192, |X|^power with and without the power == 1 short
    cut, one and two front-end waves per chunk) against the oracle: summary spectra 1e-9, per-frame chroma 1e-5, on a
    clip with a ragged last frame."""
    from oracle import iterative_f0 as o_if0
    rng = np.random.default_rng(1000 + frame_size + channels)
    n = 2 * frame_size + frame_size // 3 + 5
    x = _poly(rng, n) + (1e-3 * rng.standard_normal(n)).astype(np.float32)
    kw = dict(frame_size=frame_size, power=power, channels=channels)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wper, wut = o_if0.iterative_f0_frames(x, FS, **kw)
    ut = eng.iterative_f0_spectra(x, FS, **kw)
    tot, per = eng.iterative_f0(x, FS, return_frames=True, **kw)
    assert ut.shape == wut.shape == (3, 2 * frame_size)
    np.testing.assert_allclose(ut, wut, rtol=1e-9, atol=1e-9 * np.abs(wut).max())
    np.testing.assert_allclose(per, wper, rtol=1e-5, atol=1e-300)
    np.testing.assert_allclose(tot, wper.sum(0), rtol=1e-5, atol=1e-300)


@pytest.mark.parametrize("channels", [5, 33, 64, 70, 80])
def test_channel_counts_and_packed_leftover_waves(eng, clips, channels):
    """channels % 64 leftover channels of several chunks share one front-end wave (lane -> (chunk, channel)): every
    grouping -- all leftovers (5, 33), none (64), the default 64 + 6, four leftover sets per wave (64 + 16) -- against the
    oracle's summary spectra, for a batch whose clips differ in length (chunks of unequal length never share a wave)."""
    from oracle import iterative_f0 as o_if0
    batch = [clips["poly_seed1"][:20000], clips["poly_seed2"][:20000], clips["poly_seed1"][:20000], clips["tone_E4"][:9000],
             clips["poly_seed2"][:20000]]
    got = eng.iterative_f0_batch(batch, FS, channels=channels)
    for i in (0, 1, 3):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            per, Ut = o_if0.iterative_f0_frames(batch[i], FS, channels=channels)
        ut = eng.iterative_f0_spectra(batch[i], FS, channels=channels)
        np.testing.assert_allclose(ut, Ut, rtol=1e-9, atol=1e-9 * np.abs(Ut).max())
        np.testing.assert_allclose(got[i], per.sum(0), rtol=1e-5)
    np.testing.assert_array_equal(got[0], got[2])
    np.testing.assert_array_equal(got[1], got[4])
    np.testing.assert_array_equal(got[0], eng.iterative_f0(batch[0], FS, channels=channels))


def test_device_resident_entry_equals_host_entry(eng, clips):
    """mpx_iterative_f0_dev: the stream is read where it lies in HBM, results stay there, the caller's stream carries
    the kernels.  Bit-equal to the host entry (same kernels, same order), per frame and summed, both spellings; a
    sub-frame tail, a clip shorter than one frame and the empty stream included."""
    import torch
    side = torch.cuda.Stream(device="cuda:0")
    for name, mode in (("poly_seed1", "unicode"), ("piano_like_Cmaj", "ascii"), ("short_ragged", "unicode")):
        x = clips[name]
        want_sum, want_frames = eng.iterative_f0(x, FS, return_frames=True, note_names=mode)
        xd = torch.from_numpy(x).to("cuda:0")
        d_frames = torch.full((want_frames.shape[0], 12), -1.0, dtype=torch.float64, device="cuda:0")
        d_sum = torch.full((12,), -1.0, dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        eng.iterative_f0_dev(xd.data_ptr(), x.shape[0], FS, d_frames.data_ptr(), d_sum.data_ptr(), stream=side.cuda_stream,
                             note_names=mode)
        side.synchronize()
        np.testing.assert_array_equal(d_frames.cpu().numpy(), want_frames)
        np.testing.assert_array_equal(d_sum.cpu().numpy(), want_sum)
        d_sum.fill_(-1.0)
        torch.cuda.synchronize()
        eng.iterative_f0_dev(xd.data_ptr(), x.shape[0], FS, None, d_sum.data_ptr(), note_names=mode)   # context's stream
        eng.synchronize()
        np.testing.assert_array_equal(d_sum.cpu().numpy(), want_sum)
    d_sum = torch.full((12,), -1.0, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    eng.iterative_f0_dev(None, 0, FS, None, d_sum.data_ptr())
    eng.synchronize()
    assert not d_sum.cpu().numpy().any()
    with pytest.raises(Exception):
        eng.iterative_f0_dev(xd.data_ptr(), 100, FS, None, None)
