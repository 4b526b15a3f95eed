"""CPU: the NumPy oracle against fixtures produced by the reference's own code
(tests/golden/make_golden.py).  `ref-code` fixtures pin the oracle; the
`ref-code+stub` ones (ESACF stages a7-a9) are labelled parity-unpinned."""
import json
import os
import warnings

import numpy as np
import pytest

from oracle import chromagram as o_chroma
from oracle import dsp as o_dsp
from oracle import esacf as o_esacf
from oracle import harmonic_energy as o_he
from oracle import thirdparty as tp

FS = 22050
SPELLINGS = (("unicode", ""), ("ascii", "_ascii"))   # note_names mode, fixture key suffix (make_golden.py)


@pytest.fixture(scope="module")
def clips(golden_dir):
    d = np.load(os.path.join(golden_dir, "clips.npz"))
    return {k: d[k] for k in d.files if k != "fs"}


@pytest.fixture(scope="module")
def consts(golden_dir):
    with open(os.path.join(golden_dir, "constants.json")) as fh:
        return json.load(fh)


def test_constants(consts):
    assert consts["provenance"] == "ref-code"
    for fs in (22050, 44100):
        c = consts[str(fs)]
        assert abs(float(o_dsp.bark_warp_coef(fs)) - c["bark_a"]) < 1e-15
        np.testing.assert_allclose(o_dsp.warped_remez_coefs(fs), c["remez"], rtol=0, atol=1e-15)
        for btype, key in (("low", "butter_lp_1k"), ("high", "butter_hp_1k")):
            b, a = o_dsp.butter2(1000.0, fs, btype)
            np.testing.assert_allclose(b, c[key][0], rtol=1e-13)
            np.testing.assert_allclose(a, c[key][1], rtol=1e-13)
    np.testing.assert_allclose(tp.cqt_frequencies(12, tp.note_to_hz("C3")), consts["notes_C3"], rtol=1e-15)
    np.testing.assert_allclose(o_dsp.hamming_sym(8192)[:4], consts["hamming_8192_head"], rtol=1e-13)
    assert abs(tp.note_to_hz("C3") - 130.8127826502993) < 1e-12


def test_pack_and_key_cases(consts):
    for case in consts["pack_key_cases"]:
        assert o_chroma.pack(case["chroma"]) == case["pack"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert o_chroma.detect_key(np.asarray(case["chroma"], dtype=float)) == case["key"]


def test_detect_key_reference_known_answers():
    # inputs/outputs of the reference's tests/test_key_detection.py:9-64
    cases = {
        "Cmaj": [100.0, 0, 0, 0, 100.0, 0, 0, 100.0, 0, 0, 0, 0],
        "Cmin": [50.0, 0, 50.0, 50.0, 0, 0, 0, 10.0, 0, 0, 0, 0],
        "G#maj": [0, 10.0, 0, 10.0, 0, 0, 0, 0, 10.0, 0, 10.0, 0],
    }
    for key, v in cases.items():
        assert o_chroma.detect_key(np.asarray(v)) == key
    with pytest.raises(ValueError):
        o_chroma.detect_key(np.zeros(11))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert o_chroma.detect_key(np.ones(12)) == "Cmajmin"  # quirk A.16


def test_frame_matrix_matches_reference_semantics():
    x = np.arange(10, dtype=np.float32)
    fr = o_dsp.frame_matrix(x, 4)
    assert fr.dtype == np.float64 and fr.shape == (3, 4)
    np.testing.assert_array_equal(fr[2], [8, 9, 0, 0])
    with pytest.raises(ValueError):
        o_dsp.frame_matrix(np.zeros((2, 2)), 2)
    # overlapped extension: frame f starts at f*hop
    fr = o_dsp.frame_matrix(x, 4, 2)
    assert fr.shape == (4, 4)
    np.testing.assert_array_equal(fr[3], [6, 7, 8, 9])
    assert o_dsp.frame_matrix(np.zeros(0, dtype=np.float32), 4).shape == (0, 4)


def test_esacf_stages(golden_dir):
    d = np.load(os.path.join(golden_dir, "esacf_stages.npz"))
    assert str(d["provenance"]) == "ref-code"
    names = sorted({k.split("/")[0] for k in d.files if "/" in k})
    assert len(names) >= 6
    for name in names:
        fs = 44100 if name.startswith("fs44100") else FS
        frames = d[name + "/frames"]
        w, lo, hi = o_esacf.band_split(frames, fs)
        np.testing.assert_allclose(w, d[name + "/wfir"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(lo, d[name + "/x_lo"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(hi, d[name + "/x_hi"], rtol=0, atol=1e-12)
        s = o_esacf.sacf(d[name + "/x_lo"], d[name + "/x_hi"])
        assert s.shape[-1] == (frames.shape[-1] - 1) // 2
        np.testing.assert_allclose(s, d[name + "/sacf"], rtol=0, atol=1e-11)


def test_harmonic_energy(golden_dir, clips):
    d = np.load(os.path.join(golden_dir, "harmonic_energy.npz"))
    assert str(d["provenance"]) == "ref-code"
    for name, x in clips.items():
        per = o_he.he_frames(x, FS)
        np.testing.assert_allclose(per, d[name + "/frames"], rtol=1e-12)
        total = o_he.he_compute(x, FS)
        np.testing.assert_allclose(total, d[name + "/sum"], rtol=1e-12)
        assert o_chroma.pack(total) == str(d[name + "/repr"])
        assert o_chroma.detect_key(total) == str(d[name + "/key"])
    xT = d["T_fs44100_N4096_hop1024/x"]
    np.testing.assert_allclose(o_he.he_frames(xT, 44100, 4096, 1024),
                               d["T_fs44100_N4096_hop1024/frames"], rtol=1e-12)
    np.testing.assert_allclose(
        o_he.he_compute(clips["poly_seed2"], FS, 2048, num_harmonic=3, num_octave=3, num_bins=1),
        d["kwargs_N2048_h3_o3_b1/sum"], rtol=1e-12)


def test_harmonic_energy_windows_below_bin_zero_wrap_like_python_indexing(golden_dir, clips):
    """harmonic_energy.py:53-61: `x_dft[k]` with a negative k wraps to the top of the spectrum (large num_bins).  Fixture
    made by the reference's own code (make_golden.py he_wrap); also dft_maxes' (k0, best_ind, k1) of frame 0."""
    d = np.load(os.path.join(golden_dir, "harmonic_energy_wrap.npz"))
    assert str(d["provenance"]) == "ref-code"
    keys = sorted(k[:-7] for k in d.files if k.endswith("/kwargs"))
    assert len(keys) == 5
    for key in keys:
        kw = json.loads(str(d[key + "/kwargs"]))
        x = clips[str(d[key + "/clip"])][:int(d[key + "/n"])]
        n = kw.pop("frame_size")
        np.testing.assert_allclose(o_he.he_frames(x, FS, n, **kw), d[key + "/frames"], rtol=1e-12)
        np.testing.assert_allclose(o_he.he_compute(x, FS, n, **kw), d[key + "/sum"], rtol=1e-12)
        k0, k1, _ = o_he.he_windows(FS, n, **kw)
        assert k0.min() < 0
        np.testing.assert_array_equal(d[key + "/dft_maxes_frame0"][:, 0], k0.reshape(-1))
        np.testing.assert_array_equal(d[key + "/dft_maxes_frame0"][:, 2], k1.reshape(-1))
        np.testing.assert_array_equal(o_he.he_argmax(x[:n], FS, n, **kw)[0], d[key + "/dft_maxes_frame0"][:, 1])
    with pytest.raises(IndexError):   # past the last bin: the reference's x_dft[k] raises
        o_he.he_frames(clips["poly_seed1"][:1024], FS, 1024, num_bins=200)


def test_esacf_per_frame_fixture(golden_dir, clips):
    """ESACF frame by frame against the reference run on one frame at a time (make_golden.py esacf_frames; ref-code+stub):
    poly_seed2 holds the one ill-conditioned frame of the golden clips, so its SUM cannot separate a real regression from
    the fit's chaos -- its frames can.  Both spellings."""
    d = np.load(os.path.join(golden_dir, "esacf_frames.npz"))
    e = np.load(os.path.join(golden_dir, "esacf_e2e.npz"))
    for mode, sfx in SPELLINGS:
        for name in ("poly_seed2", "poly_seed1", "piano_like_Cmaj"):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                per = o_esacf.esacf_frames(clips[name], FS, frame_size=1023, note_names=mode)
            np.testing.assert_allclose(per, d[name + "/frames" + sfx], rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(d[name + "/frames" + sfx].sum(0), e[name + "/sum" + sfx], rtol=1e-12, atol=1e-12)


def test_he_windows_T_shape():
    k0, k1, w = o_he.he_windows(44100, 4096)
    assert k0.min() == 47 and k1.max() == 371  # k' in [49, 367], +-2h bins
    assert k0.shape == (12, 4)
    np.testing.assert_array_equal(w[0], [1.0, 0.5, 1.0, 0.5])


def test_esacf_end_to_end_unpinned(golden_dir, clips):
    """ref-code+stub: the reference's esacf.py driving our stand-ins for
    librosa.effects.time_stretch / peakutils (real scipy curve_fit inside)."""
    d = np.load(os.path.join(golden_dir, "esacf_e2e.npz"))
    assert str(d["provenance"]) == "ref-code+stub"
    for mode, sfx in SPELLINGS:
        for name in ("tone_Csharp3", "tone_E4", "tones_G2_B2_Gsharp3", "piano_like_Cmaj", "short_ragged"):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                total = o_esacf.esacf_compute(clips[name], FS, note_names=mode)
            np.testing.assert_allclose(total, d[name + "/sum" + sfx], rtol=1e-6, atol=1e-9)
            assert o_chroma.pack(total) == str(d[name + "/repr" + sfx])
            assert o_chroma.detect_key(total) == str(d[name + "/key" + sfx])
            if mode == "unicode":
                assert all(total[pc] == 0.0 for pc in o_esacf.SHARP_PITCH_CLASSES)  # quirk A.18
    # the clip the reference's tests/test.py:15 expects "010000000000" for: with ASCII note names the C# bin is the
    # only non-zero one, with unicode names (librosa >= 0.8) the reference loses it
    assert str(d["tone_Csharp3/repr_ascii"]) == "020000000000" and str(d["tone_Csharp3/repr"]) == "000000000000"
    s = np.load(os.path.join(golden_dir, "esacf_stages.npz"))
    for name in ("piano_like_Cmaj", "poly_seed1"):
        for i, sac in enumerate(s[name + "/sacf"]):
            e = o_esacf.esacf_enhance(sac)
            np.testing.assert_allclose(e, d[name + "/esacf"][i], rtol=0, atol=1e-14)
            pk = tp.peak_indexes(e, 0.1, 10)
            ref_pk = d[name + "/peaks"][i]
            np.testing.assert_array_equal(pk, ref_pk[ref_pk >= 0])
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                q = tp.peak_interpolate(np.arange(e.shape[0]), e, pk)
            ref_q = d[name + "/peaks_interp"][i]
            np.testing.assert_allclose(q, ref_q[~np.isnan(ref_q)], rtol=1e-7)


def test_lmdif_matches_minpack():
    """Our MINPACK restatement vs the real thing (scipy wraps MINPACK lmdif)."""
    scipy_optimize = pytest.importorskip("scipy.optimize")
    rng = np.random.default_rng(5)
    for _ in range(20):
        c = rng.uniform(300, 320)
        xs = np.arange(int(c) - 10, int(c) + 11, dtype=float)
        ys = rng.uniform(0.05, 2) * np.exp(-((xs - c) ** 2) / (2 * rng.uniform(2, 30) ** 2)) + 1e-3 * rng.standard_normal(21)
        p0 = [float(ys.max()), float(xs[0]), 5.0]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            popt, _, info, _, ier = scipy_optimize.curve_fit(tp.gaussian, xs, ys, p0, full_output=True)
            p, mine_info, nfev = tp.lmdif(lambda q: tp.gaussian(xs, *q) - ys, p0)
        assert mine_info == ier
        np.testing.assert_allclose(p, popt, rtol=1e-6)


def test_time_stretch_truncation_identity():
    rng = np.random.default_rng(3)
    y = np.clip(rng.standard_normal(511), 0, None)
    for r in range(2, 7):
        out = tp.time_stretch(y, r)
        assert out.shape[0] == int(round(511 / r))
        np.testing.assert_allclose(out, y[:out.shape[0]], atol=1e-14)
    assert tp.time_stretch_is_truncation(1022) and not tp.time_stretch_is_truncation(2047)


def test_prime_multif0(golden_dir, clips):
    """Method 4 (next-tier row f2): pinned by the reference's code + the real matplotlib.mlab."""
    from oracle import prime_multif0 as o_prime
    d = np.load(os.path.join(golden_dir, "prime_multif0.npz"))
    assert str(d["provenance"]) == "ref-code+notes"
    for mode, sfx in SPELLINGS:
        for name, x in clips.items():
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                total = o_prime.prime_compute(x, FS, note_names=mode)
            np.testing.assert_allclose(total, d[name + "/sum" + sfx], rtol=1e-12, atol=0)
            assert o_chroma.pack(total) == str(d[name + "/repr" + sfx])
            assert o_chroma.detect_key(total) == str(d[name + "/key" + sfx])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            np.testing.assert_allclose(o_prime.prime_compute(clips["poly_seed1"], FS, 2, 3, 3, 3, note_names=mode),
                                       d["kwargs_h2_o3_e3_r3/sum" + sfx], rtol=1e-12)
    assert not np.array_equal(d["tone_Csharp3/sum"], d["tone_Csharp3/sum_ascii"])
    ws = [w for _, w in o_prime.candidates(FS)]
    assert min(ws) == 357 and max(ws) == 1348 and len(ws) == 24


def test_iterative_f0(golden_dir, clips):
    """Method 3 (next-tier row f1): pinned by the reference's own code (only closed-form librosa helpers
    stood in)."""
    from oracle import iterative_f0 as o_if0
    d = np.load(os.path.join(golden_dir, "iterative_f0.npz"))
    assert str(d["provenance"]) == "ref-code+notes"
    for name in ("tone_E4", "short_ragged"):   # two clips keep the CPU suite short; the GPU suite covers all eight
        for mode, sfx in SPELLINGS:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                per, Ut = o_if0.iterative_f0_frames(clips[name], FS, note_names=mode)
            np.testing.assert_allclose(per, d[name + "/frames" + sfx], rtol=1e-12, atol=0)
            np.testing.assert_allclose(per.sum(0), d[name + "/sum" + sfx], rtol=1e-12, atol=0)
            np.testing.assert_allclose(Ut[0][:512], d[name + "/ut0_head"], rtol=1e-12)
            assert o_chroma.pack(per.sum(0)) == str(d[name + "/repr" + sfx])
    fc = o_if0.channel_frequencies()
    assert len(fc) == 70 and 64 < fc[0] < 65 and 5000 < fc[-1] < 5100
