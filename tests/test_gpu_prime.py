"""GPU parity: Prime-multiF0 (reference method 4) through the C ABI vs fixtures made by
the reference's own code (with the real matplotlib.mlab) and vs the oracle.  fp64; north_star bar 1e-5,
held here at 1e-9."""
import json
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FS = 22050
RTOL = 1e-9
SPELLINGS = (("unicode", ""), ("ascii", "_ascii"))   # note_names mode, fixture key suffix (make_golden.py)


@pytest.fixture(scope="module")
def eng():
    import chord_detection_amd as cd
    return cd.get_engine(0)


@pytest.fixture(scope="module")
def clips(golden_dir):
    d = np.load(os.path.join(golden_dir, "clips.npz"))
    return {k: d[k] for k in d.files if k != "fs"}


def test_golden_clips_sum_string_key(eng, clips, golden_dir):
    import chord_detection_amd as cd
    d = np.load(os.path.join(golden_dir, "prime_multif0.npz"))
    assert cd.METHODS[4] is cd.MultipitchPrimeMultiF0
    assert cd.MultipitchPrimeMultiF0.display_name() == "Prime-multiF0 (Camacho, Kaver-Oreamuno)"
    expected = json.load(open(os.path.join(golden_dir, "constants.json")))["test_py_expected"]
    for mode, sfx in SPELLINGS:
        for name, x in clips.items():
            c = cd.MultipitchPrimeMultiF0((x, FS), note_names=mode).compute_pitches()
            np.testing.assert_allclose(c.as_array(), d[name + "/sum" + sfx], rtol=RTOL, atol=0)
            assert repr(c) == str(d[name + "/repr" + sfx])
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                assert c.key() == str(d[name + "/key" + sfx])
            if mode == "unicode":
                assert all(c[pc] == 0.0 for pc in (1, 3, 6, 8, 10))   # quirk A.18
            if name in expected:
                print("prime  %-8s %-20s engine %s  tests/test.py expects %s" % (mode, name, repr(c), expected[name]))
        got = eng.prime_multif0(clips["poly_seed1"], FS, 2, 3, 3, 3, note_names=mode)
        np.testing.assert_allclose(got, d["kwargs_h2_o3_e3_r3/sum" + sfx], rtol=RTOL)
    assert np.any(eng.prime_multif0(clips["tone_Csharp3"], FS, note_names="ascii")[[1, 3, 6, 8, 10]] > 0)
    got = eng.prime_multif0_batch([clips["tone_Csharp3"], clips["tone_E4"]], FS, note_names="ascii")
    np.testing.assert_allclose(got[0], d["tone_Csharp3/sum_ascii"], rtol=RTOL)
    np.testing.assert_allclose(got[1], d["tone_E4/sum_ascii"], rtol=RTOL)
    with pytest.raises(ValueError):
        eng.prime_multif0(clips["tone_E4"], FS, note_names="latin1")


def test_batch_edge_cases_vs_oracle(eng, clips):
    from oracle import prime_multif0 as o_prime
    rng = np.random.default_rng(4)
    batch = [clips["tone_E4"], clips["short_ragged"], np.zeros(0, dtype=np.float32), np.zeros(700, dtype=np.float32),
             rng.standard_normal(5000).astype(np.float32), clips["poly_seed2"][:357]]
    got = eng.prime_multif0_batch(batch, FS)
    assert got.shape == (6, 12)
    for i, x in enumerate(batch):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = o_prime.prime_compute(x, FS) if len(x) else np.zeros(12)
        # atol: a tail frame holding only 1-2 samples has a FLAT spectrum (|X[k]| equal for all k up to
        # rounding), so the reference's argmax -- and with it the pitch class that receives that ~1e-9
        # magnitude -- is decided by rounding noise.  Everything above that level must agree to 1e-9.
        np.testing.assert_allclose(got[i], want, rtol=RTOL, atol=1e-7)
    again = eng.prime_multif0_batch(batch[::-1], FS)[::-1]
    assert np.array_equal(got, again)                      # deterministic, independent of batch position
    # 44.1 kHz: the low candidates need frames of 2049..2696 samples -> the 8192-point chirp-z class
    rng = np.random.default_rng(44)
    t = np.arange(60000) / 44100.0
    x44 = (0.2 * sum(0.5 ** h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6)) for f0 in (220.0, 330.0) for h in (1, 2, 3))
           + 1e-3 * rng.standard_normal(t.shape[0])).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_allclose(eng.prime_multif0(x44, 44100), o_prime.prime_compute(x44.astype(np.float64), 44100),
                                   rtol=RTOL, atol=1e-7)
    # 96 kHz: frames of 1555..5871 samples; the chirp-z spans N + N/4 points, so they fit the 8192-point class
    t = np.arange(40000) / 96000.0
    x96 = (0.3 * np.sin(2 * np.pi * 261.63 * t) + 0.2 * np.sin(2 * np.pi * 392.0 * t) + 1e-3 * rng.standard_normal(t.shape[0])).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_allclose(eng.prime_multif0(x96, 96000), o_prime.prime_compute(x96.astype(np.float64), 96000),
                                   rtol=RTOL, atol=1e-7)
    # above ~107 kHz the lowest candidates' frames exceed 6553 samples: the input decimated by R, R passes of the 8192-point
    # chirp-z (192 kHz: frames of 3110..11743 samples, R up to 3; 120 kHz: up to 7339 samples, R = 2; 260 kHz: up to 15901, R = 4)
    for fsr, nsmp in ((192000, 70000), (120000, 40000), (260000, 50000)):
        tt = np.arange(nsmp) / float(fsr)
        xx = (0.3 * np.sin(2 * np.pi * 261.63 * tt) + 0.2 * np.sin(2 * np.pi * 392.0 * tt) + 0.1 * np.sin(2 * np.pi * 164.81 * tt)
              + 1e-3 * rng.standard_normal(nsmp)).astype(np.float32)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            np.testing.assert_allclose(eng.prime_multif0(xx, fsr), o_prime.prime_compute(xx.astype(np.float64), fsr), rtol=RTOL, atol=1e-7)
    with pytest.raises(NotImplementedError):
        eng.prime_multif0(np.zeros(100, dtype=np.float32), 384000)   # 8/f*fs > 16384 samples for the low candidates
    with pytest.raises(ValueError):
        eng.prime_multif0(np.zeros((2, 2), dtype=np.float32), FS)


def test_frame_pairs_silence_and_one_frame_mode(eng):
    """Round 3: prime_pers_kernel transforms two consecutive real frames of a clip at once and separates them by conjugate
    symmetry.  (a) A frame of digital silence next to a loud one must contribute the reference's exact nothing (its spectrum
    is all zeros: the argmax lands on the DC bin, hz_to_note raises, `continue`), not the partner's rounding noise;
    (b) clips whose frame counts per candidate are odd, even and 1; (c) 48 kHz: the lowest candidates (2731..3277
    samples) run one frame per transform on the same kernel, the others in pairs."""
    from oracle import prime_multif0 as o_prime
    rng = np.random.default_rng(7)
    t = np.arange(9000) / FS
    tone = (0.5 * np.sin(2 * np.pi * 329.63 * t) + 0.25 * np.sin(2 * np.pi * 659.26 * t)).astype(np.float32)
    cases = {
        "tone_then_silence": np.concatenate([tone, np.zeros(9000, dtype=np.float32)]),
        "silence_then_tone": np.concatenate([np.zeros(9000, dtype=np.float32), tone]),
        "alternating": np.concatenate([tone[:1348], np.zeros(1348, dtype=np.float32)] * 4),
        "one_frame_each": tone[:300],
        "exactly_two_frames_of_the_longest": tone[:2 * 1348],
        "all_silence": np.zeros(5000, dtype=np.float32),
    }
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, x in cases.items():
            for mode in ("unicode", "ascii"):
                got = eng.prime_multif0(x, FS, note_names=mode)
                want = o_prime.prime_compute(x.astype(np.float64), FS, note_names=mode)
                np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-12, err_msg=name)
                if name == "all_silence":
                    assert not got.any()
        # silence contributes NOTHING: the clip with silence appended equals the clip alone wherever the frames line up
        # (candidate by candidate the tone's frames are the same samples; the extra frames are silent)
        a = eng.prime_multif0(cases["tone_then_silence"], FS, note_names="ascii")
        b = o_prime.prime_compute(cases["tone_then_silence"].astype(np.float64), FS, note_names="ascii")
        assert np.array_equal(a == 0, b == 0)   # exactly the oracle's empty pitch classes: no 1e-17 leak from a loud partner
        # 48 kHz, a chord with noise, against the oracle
        t48 = np.arange(30000) / 48000.0
        x48 = (0.3 * np.sin(2 * np.pi * 196.0 * t48) + 0.2 * np.sin(2 * np.pi * 246.94 * t48) + 0.1 * np.sin(2 * np.pi * 392.0 * t48)
               + 1e-3 * rng.standard_normal(t48.shape[0])).astype(np.float32)
        np.testing.assert_allclose(eng.prime_multif0(x48, 48000), o_prime.prime_compute(x48.astype(np.float64), 48000),
                                   rtol=RTOL, atol=1e-7)
        batch = eng.prime_multif0_batch([x48, x48[:12345], np.zeros(4000, dtype=np.float32)], 48000)
        np.testing.assert_array_equal(batch[0], eng.prime_multif0(x48, 48000))
        assert not batch[2].any()


def test_equal_length_clips_share_one_item_list(eng, clips):
    """A batch of equal-length clips takes the path where the host builds ONE clip's item list and the kernel derives
    the others: every clip must come out exactly as on its own, and as in a ragged batch (the general path)."""
    names = ["poly_seed1", "poly_seed2", "tone_E4", "piano_like_Cmaj", "poly_seed1"]
    same = np.stack([clips[n][:30000] for n in names])
    got = eng.prime_multif0_batch(same, FS)
    assert got.shape == (5, 12) and np.abs(got).sum() > 0
    for i in range(5):
        np.testing.assert_array_equal(got[i], eng.prime_multif0(same[i], FS))
    ragged = eng.prime_multif0_batch([same[0], same[1][:29999], same[2]], FS)
    np.testing.assert_array_equal(ragged[0], got[0])
    np.testing.assert_array_equal(ragged[2], got[2])
    np.testing.assert_array_equal(got[0], got[4])
    two = eng.prime_multif0_batch(np.zeros((3, 0), dtype=np.float32), FS)      # equal length zero
    assert two.shape == (3, 12) and np.all(two == 0)


def test_device_resident_entry_equals_host_entry(eng, clips):
    """mpx_prime_multif0_dev: signal and result in HBM, kernels on the caller's stream; bit-equal to the host entry."""
    import torch
    side = torch.cuda.Stream(device="cuda:0")
    for name, mode in (("poly_seed1", "unicode"), ("piano_like_Cmaj", "ascii"), ("short_ragged", "unicode")):
        x = clips[name]
        want = eng.prime_multif0(x, FS, note_names=mode)
        xd = torch.from_numpy(x).to("cuda:0")
        d_sum = torch.full((12,), -1.0, dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        eng.prime_multif0_dev(xd.data_ptr(), x.shape[0], FS, d_sum.data_ptr(), stream=side.cuda_stream, note_names=mode)
        side.synchronize()
        np.testing.assert_array_equal(d_sum.cpu().numpy(), want)
        d_sum.fill_(-1.0)
        torch.cuda.synchronize()
        eng.prime_multif0_dev(xd.data_ptr(), x.shape[0], FS, d_sum.data_ptr(), note_names=mode)
        eng.synchronize()
        np.testing.assert_array_equal(d_sum.cpu().numpy(), want)
    d_sum = torch.full((12,), -1.0, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    eng.prime_multif0_dev(None, 0, FS, d_sum.data_ptr())
    eng.synchronize()
    assert not d_sum.cpu().numpy().any()
    with pytest.raises(Exception):
        eng.prime_multif0_dev(xd.data_ptr(), 100, FS, None)


@pytest.mark.parametrize("drop_db", [80, 120, 200])
def test_paired_frames_with_a_large_level_difference(eng, drop_db):
    """prime_pers_kernel transforms two consecutive frames of a candidate as a + i b and separates their spectra by
    conjugate symmetry: the quiet one of a pair carries its loud partner's rounding noise, ~1e-16 of the LOUD frame's
    magnitudes.  A clip whose second half is 80 / 120 / 200 dB below its first (a decay tail, a fade-out): the quiet frames'
    arg-max may differ from numpy's per-frame FFT once that noise reaches their own level, but what they add to the clip's
    chroma is bounded by that noise -- the sums agree with the oracle to 1e-5 relative plus 1e-13 of the largest bin (the
    bound include/mpx.h states), with both note spellings."""
    from oracle import prime_multif0 as o_prime
    rng = np.random.default_rng(drop_db)
    n = 2 * FS
    t = np.arange(n) / FS
    x = np.zeros(n)
    for f0 in (196.0, 246.94, 329.63):
        for h in range(1, 6):
            x += 0.6 ** h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6))
    x = 0.25 * x + 0.01 * rng.standard_normal(n)
    env = np.ones(n)
    env[n // 2:] = 10.0 ** (-drop_db / 20.0)                       # an abrupt drop: pairs straddle it for every candidate
    env[n // 4:n // 2] = np.linspace(1.0, 10.0 ** (-drop_db / 40.0), n // 2 - n // 4)   # and a ramp: every ratio in between
    x = (x * env).astype(np.float32)
    for mode in ("unicode", "ascii"):
        got = eng.prime_multif0(x, FS, note_names=mode)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = o_prime.prime_compute(x, FS, note_names=mode)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-13 * np.abs(want).max())
        assert np.abs(want).max() > 0


@pytest.mark.parametrize("fs,kw", [(8000, dict(num_octave=3)), (16000, dict(num_harmonic=2, num_octave=2)),
                                   (11025, dict(harmonic_multiples_elim=3, harmonic_elim_runs=3)), (32000, dict(num_octave=1))])
def test_wave_kernel_classes_at_other_rates_and_parameters(eng, fs, kw):
    """Round 5: prime_wave_kernel runs every candidate whose chirp-z fits 1024 or 2048 points (a wave per pair of frames, both
    transforms in registers): short frames at low rates and higher octaves, more elimination rounds, single clips and batches
    of equal and of ragged lengths -- against the oracle and against each other."""
    from oracle import prime_multif0 as o_prime
    rng = np.random.default_rng(fs)
    n = int(1.3 * fs)
    t = np.arange(n) / fs
    xs = []
    for f0s in ((196.0, 246.94), (261.63, 329.63, 392.0), (146.83,)):
        x = sum(0.6 ** h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6)) for f0 in f0s for h in (1, 2, 3) if f0 * h < fs / 2)
        xs.append((0.3 * x + 1e-3 * rng.standard_normal(n)).astype(np.float32))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = [o_prime.prime_compute(x.astype(np.float64), fs, **kw) for x in xs]
    same = eng.prime_multif0_batch(np.stack(xs), fs, **kw)                     # equal lengths: items from (clip, pair)
    ragged = eng.prime_multif0_batch([xs[0], xs[1][: n - 777], xs[2]], fs, **kw)   # the general item list
    for i, x in enumerate(xs):
        np.testing.assert_allclose(same[i], want[i], rtol=1e-7, atol=1e-6)
        np.testing.assert_array_equal(same[i], eng.prime_multif0(x, fs, **kw))
    np.testing.assert_array_equal(ragged[0], same[0])
    np.testing.assert_array_equal(ragged[2], same[2])
