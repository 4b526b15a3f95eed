#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE'S OWN CODE.

Runs only in the authoring container, where the reference is mounted read-only
at /root/reference (it does not exist on the GPU box; tests only read the
committed .npz/.json files).  Nothing of the reference is copied: this script
imports it, feeds it inputs, and stores inputs + outputs.

Import recipe (the reference cannot be imported as-is here: librosa, peakutils,
soundfile, numba are not installed and scipy.signal.hamming was removed from
SciPy >= 1.13):
  * sys.modules stand-ins for librosa / peakutils / soundfile / numba built from
    oracle/thirdparty.py (our restatement of those libraries' published
    behaviour); peakutils' gaussian fit calls the REAL scipy.optimize.curve_fit,
    exactly as peakutils does;
  * alias scipy.signal.hamming = scipy.signal.windows.hamming.

Provenance labels stored with every array:
  ref-code       reference code + numpy/scipy only           (pins the oracle)
  ref-code+stub  reference code calling our librosa.effects.time_stretch /
                 peakutils stand-ins (ESACF stages a7-a9)    (parity UNPINNED)
  ref-code+notes reference code + numpy/scipy/matplotlib, with a stand-in for
                 librosa.hz_to_note (closed form) whose SPELLING of sharps is a
                 choice: librosa >= 0.8 writes "C♯", librosa < 0.8 wrote "C#".
                 The reference's Chromagram loses unicode sharps on `+=`
                 (chromagram.py:19-29), so methods 1, 3 and 4 have TWO sets of
                 expected values: keys `<clip>/sum` ... are the unicode
                 spelling, `<clip>/sum_ascii` ... the ASCII one.

Usage:  python tests/golden/make_golden.py            (every fixture)
        python tests/golden/make_golden.py he_wrap esacf_frames   (only the named round-5 sections; the others stay as committed)
"""
import json
import os
import sys
import types
import warnings

os.environ["MPLBACKEND"] = "Agg"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import scipy.optimize
import scipy.signal
import scipy.signal.windows

from oracle import thirdparty as tp

REFERENCE = "/root/reference"
_CLIPS = {}  # path -> (float32 array, fs) served by the librosa.load stand-in


def install_stubs():
    scipy.signal.hamming = scipy.signal.windows.hamming

    librosa = types.ModuleType("librosa")
    librosa.load = lambda path, *a, **k: (_CLIPS[str(path)][0].copy(), _CLIPS[str(path)][1])
    librosa.hz_to_note = tp.hz_to_note   # unicode spelling; set_spelling() swaps it
    librosa.note_to_hz = tp.note_to_hz
    librosa.cqt_frequencies = tp.cqt_frequencies
    librosa.tone = tp.tone
    effects = types.ModuleType("librosa.effects")
    effects.time_stretch = lambda y, rate: tp.time_stretch(y, rate)
    librosa.effects = effects

    peakutils = types.ModuleType("peakutils")
    peakutils.indexes = lambda y, thres=0.3, min_dist=1: tp.peak_indexes(y, thres, min_dist)

    def gaussian_fit(x, y):
        if len(x) < 3:
            raise RuntimeError("At least 3 points required for Gaussian fitting")
        initial = [np.max(y), x[0], (x[1] - x[0]) * 5]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            params, _ = scipy.optimize.curve_fit(tp.gaussian, x, y, initial)
        return params[1]

    def interpolate(x, y, ind=None, width=10):
        out = []
        for i in ind:
            sl = slice(i - width, i + width + 1)
            try:
                out.append(gaussian_fit(x[sl], y[sl]))
            except Exception:
                pass
        return np.array(out)

    peakutils.interpolate = interpolate
    numba = types.ModuleType("numba")
    numba.njit = numba.jit = lambda f=None, **k: f
    soundfile = types.ModuleType("soundfile")
    for name, mod in (("librosa", librosa), ("librosa.effects", effects), ("peakutils", peakutils),
                      ("numba", numba), ("soundfile", soundfile)):
        sys.modules[name] = mod
    sys.path.insert(0, REFERENCE)


def set_spelling(mode):
    """Which librosa the stand-in imitates: 'unicode' (>= 0.8) or 'ascii' (< 0.8).  The reference looks
    `librosa.hz_to_note` up at call time, so swapping the module attribute is enough."""
    uni = {"unicode": True, "ascii": False}[mode]
    sys.modules["librosa"].hz_to_note = lambda f, octave=False: tp.hz_to_note(f, octave=octave, unicode=uni)


SPELLINGS = (("unicode", ""), ("ascii", "_ascii"))   # (mode, key suffix)


# ---------------------------------------------------------------- inputs (G1)
def tone_clip(freqs, sr=22050, length=44100):
    t = np.zeros(length)
    for f in freqs:
        t += tp.tone(f, sr=sr, length=length)
    return t.astype(np.float32)


def piano_like(sr=22050, length=44100, notes=(261.63, 329.63, 392.0)):
    n = np.arange(length) / sr
    y = np.zeros(length)
    for f0 in notes:
        for h in range(1, 9):
            y += (0.6 ** (h - 1)) * np.exp(-1.5 * n * h ** 0.5) * np.sin(2 * np.pi * f0 * h * n)
    return (0.9 * y / np.max(np.abs(y))).astype(np.float32)


def poly_clip(seed, sr=22050, length=44100):
    rng = np.random.default_rng(seed)
    n = np.arange(length) / sr
    y = np.zeros(length)
    for _ in range(int(rng.integers(2, 5))):
        f0 = 440.0 * 2.0 ** ((int(rng.integers(36, 85)) - 69) / 12.0)
        ph = rng.uniform(0, 2 * np.pi)
        for h in range(1, 9):
            if f0 * h < sr / 2:
                y += (0.7 ** (h - 1)) * np.sin(2 * np.pi * f0 * h * n + ph * h)
    y += 0.01 * rng.standard_normal(length)
    return (0.9 * y / np.max(np.abs(y))).astype(np.float32)


def make_clips():
    # note sets of the reference's tests/gen_test_clips.py:13-43 (float arrays, no WAV round trip)
    clips = {
        "tone_Csharp3": tone_clip([138.59]),
        "tone_E4": tone_clip([329.63]),
        "tones_E2_F3": tone_clip([82.41, 174.61]),
        "tones_G3_Asharp4": tone_clip([196, 466.16]),
        "tones_G2_B2_Gsharp3": tone_clip([98, 123.47, 207.65]),
        "piano_like_Cmaj": piano_like(),
        "poly_seed1": poly_clip(1),
        "poly_seed2": poly_clip(2),
        "short_ragged": poly_clip(3)[:3000],   # shorter than one HE frame, ragged ESACF tail
    }
    return clips


# ---------------------------------------------------------------- round 5 sections (files of their own)
def golden_he_wrap(chord_detection, clips, fs):
    """harmonic_energy.py:53-61 with windows that start BELOW bin 0: `x_dft[k]` with a negative k is plain Python indexing and
    wraps to the top of the spectrum.  Reached with large `num_bins` (k' - num_bins * harmonic < 0).  ref-code."""
    from oracle import harmonic_energy as o_he
    he = {"provenance": np.array("ref-code")}
    cases = [("N1024_b30", dict(frame_size=1024, num_bins=30), "poly_seed1"),
             ("N4096_b110_h2_o2", dict(frame_size=4096, num_bins=110), "piano_like_Cmaj"),
             ("N8192_b200", dict(frame_size=8192, num_bins=200), "poly_seed2"),
             ("N2048_b55_h3_o1", dict(frame_size=2048, num_bins=55, num_harmonic=3, num_octave=1), "tones_G2_B2_Gsharp3"),
             ("N1000_b27", dict(frame_size=1000, num_bins=27), "short_ragged")]
    for key, kw, clip in cases:
        x = clips[clip][:4 * kw["frame_size"] + 77]
        _CLIPS["__wrap__"] = (x, fs)
        obj = chord_detection.MultipitchHarmonicEnergy("__wrap__", **kw)
        total = obj.compute_pitches()
        assert min(k0 for k0, _, _ in obj.dft_maxes) < 0, key          # the case does wrap
        he[key + "/clip"] = np.array(clip)
        he[key + "/n"] = np.array(x.shape[0])
        he[key + "/kwargs"] = np.array(json.dumps(kw))
        he[key + "/sum"] = np.array([total[i] for i in range(12)])
        F = int(np.ceil(len(x) / kw["frame_size"]))
        per = []
        for f in range(F):
            _CLIPS["__frame__"] = (x[f * kw["frame_size"]:(f + 1) * kw["frame_size"]], fs)
            c = chord_detection.MultipitchHarmonicEnergy("__frame__", **kw).compute_pitches()
            per.append([c[i] for i in range(12)])
        he[key + "/frames"] = np.array(per)
        # (k0, best_ind, k1) of the first frame's windows: what MultipitchHarmonicEnergy.dft_maxes holds (harmonic_energy.py:65)
        he[key + "/dft_maxes_frame0"] = np.array(obj.dft_maxes[:len(obj.dft_maxes) // F], dtype=np.int64)
        okw = {k: v for k, v in kw.items() if k != "frame_size"}
        np.testing.assert_allclose(o_he.he_frames(x, fs, kw["frame_size"], **okw), np.array(per), rtol=1e-12, atol=0)
    np.savez_compressed(os.path.join(HERE, "harmonic_energy_wrap.npz"), **he)


def golden_esacf_frames(chord_detection, clips, fs):
    """ESACF per FRAME for the clips whose sums hold ill-conditioned fits (poly_seed2: tests/test_gpu_esacf.py): the
    reference run on one frame at a time, both note spellings.  ref-code+stub (a7-a8 go through the stand-ins)."""
    es = {"provenance": np.array("ref-code+stub")}
    N = int(fs * 46.4 / 1000)
    for mode, sfx in SPELLINGS:
        set_spelling(mode)
        for name in ("poly_seed2", "poly_seed1", "piano_like_Cmaj"):
            x = clips[name]
            F = int(np.ceil(len(x) / N))
            per = []
            for f in range(F):
                _CLIPS["__frame__"] = (x[f * N:(f + 1) * N], fs)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    c = chord_detection.MultipitchESACF("__frame__").compute_pitches()
                per.append([c[i] for i in range(12)])
            es[name + "/frames" + sfx] = np.array(per)
    set_spelling("unicode")
    np.savez_compressed(os.path.join(HERE, "esacf_frames.npz"), **es)


ROUND5 = {"he_wrap": golden_he_wrap, "esacf_frames": golden_esacf_frames}


def main():
    install_stubs()
    import chord_detection
    from chord_detection import chromagram as ref_chroma
    from chord_detection import esacf as ref_esacf
    from chord_detection.dsp import frame as ref_frame, lowpass as ref_lowpass, wfir as ref_wfir
    from oracle import chromagram as o_chroma, dsp as o_dsp, esacf as o_esacf, harmonic_energy as o_he

    assert list(chord_detection.METHODS.keys()) == [1, 2, 3, 4]
    fs = 22050
    clips = make_clips()
    for name, x in clips.items():
        _CLIPS[name] = (x, fs)
    only = [a for a in sys.argv[1:] if a in ROUND5]
    if only:   # the named sections alone; the clips are regenerated in memory (same seeds) and checked against the committed file
        committed = np.load(os.path.join(HERE, "clips.npz"))
        for name, x in clips.items():
            np.testing.assert_array_equal(committed[name], x)
        for a in only:
            ROUND5[a](chord_detection, clips, fs)
        print("golden fixtures written:", ", ".join(only))
        return
    np.savez_compressed(os.path.join(HERE, "clips.npz"), fs=fs, **clips)

    # ------------------------------------------------------ G7 constants
    consts = {"provenance": "ref-code"}
    for f in (22050, 44100):
        consts[str(f)] = {
            "bark_a": float(ref_wfir._bark_warp_coef(f)),
            "remez": [float(v) for v in ref_wfir._warped_remez_coefs(f, 12)],
            "butter_lp_1k": [[float(v) for v in p] for p in scipy.signal.butter(2, [1000 / (f / 2)], btype="low")],
            "butter_hp_1k": [[float(v) for v in p] for p in scipy.signal.butter(2, [1000 / (f / 2)], btype="high")],
        }
    consts["notes_C3"] = [float(v) for v in tp.cqt_frequencies(12, fmin=tp.note_to_hz("C3"))]
    consts["hamming_8192_head"] = [float(v) for v in scipy.signal.hamming(8192)[:4]]

    # ------------------------------------------------------ G4/G5 chromagram + key
    pack_cases = []
    rng = np.random.default_rng(7)
    vecs = [
        [5, 0, 0, 0, 2, 0, 0, 1, 0, 0, 0, 0],
        list(range(1, 13)),
        [0.0] * 12,
        [1.0] * 12,
        [0.5, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5, 8.5, 0.25, 0.75, 9.0],
        [2.0, 3.0, 5.0, 7.0, 11.0, 13.0, 17.0, 19.0, 23.0, 29.0, 31.0, 37.0],
        [100.0, 0, 0, 0, 100.0, 0, 0, 100.0, 0, 0, 0, 0],
        [50.0, 0, 50.0, 50.0, 0, 0, 0, 10.0, 0, 0, 0, 0],
        [0, 10.0, 0, 10.0, 0, 0, 0, 0, 10.0, 0, 10.0, 0],
    ] + [list(rng.uniform(0.1, 50, 12)) for _ in range(8)]
    for v in vecs:
        c = ref_chroma.Chromagram()
        for i, val in enumerate(v):
            c[i] = float(val)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            key = c.key()
        pack_cases.append({"chroma": [float(t) for t in v], "pack": repr(c), "key": key})
        assert o_chroma.pack(v) == repr(c), (v, o_chroma.pack(v), repr(c))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert o_chroma.detect_key(np.asarray(v, dtype=float)) == key
    consts["pack_key_cases"] = pack_cases
    # the expectations the reference's tests/test.py:14-20 declares (and only prints) for its five tone clips:
    # data, kept so the GPU tests can print them next to the engine's strings in both note spellings
    consts["test_py_expected"] = {
        "tone_Csharp3": "010000000000", "tone_E4": "000010000000", "tones_E2_F3": "000011000000",
        "tones_G3_Asharp4": "000000010010", "tones_G2_B2_Gsharp3": "000000011001"}
    with open(os.path.join(HERE, "constants.json"), "w") as fh:
        json.dump(consts, fh, indent=1)

    # ------------------------------------------------------ G2 ESACF stages (ref-code)
    stage = {}
    N = ref_esacf.MultipitchESACF("tone_E4").ham_samples
    assert N == 1023
    for name in ("tones_G2_B2_Gsharp3", "piano_like_Cmaj", "poly_seed1", "short_ragged"):
        x = clips[name]
        frames = list(ref_frame.frame_cutter(x, N))
        pick = sorted(set([0, 1, len(frames) - 1]))
        xin, xw, xlo, xhi, xs = [], [], [], [], []
        for f in pick:
            fr = frames[f]
            w = ref_wfir.wfir(fr, fs, 12)
            hi = ref_esacf._highpass_filter(w, fs)
            hi = np.clip(hi, 0, None)
            hi = ref_lowpass.lowpass_filter(hi, fs, 1000)
            lo = ref_lowpass.lowpass_filter(w, fs, 1000)
            s = ref_esacf._sacf([lo, hi])
            xin.append(fr); xw.append(w); xlo.append(lo); xhi.append(hi); xs.append(s)
        stage[name + "/frame_idx"] = np.array(pick)
        stage[name + "/frames"] = np.array(xin)
        stage[name + "/wfir"] = np.array(xw)
        stage[name + "/x_lo"] = np.array(xlo)
        stage[name + "/x_hi"] = np.array(xhi)
        stage[name + "/sacf"] = np.array(xs)
        # oracle check at generation time
        ow, olo, ohi = o_esacf.band_split(np.array(xin), fs)
        np.testing.assert_allclose(ow, np.array(xw), rtol=0, atol=1e-12)
        np.testing.assert_allclose(olo, np.array(xlo), rtol=0, atol=1e-12)
        np.testing.assert_allclose(ohi, np.array(xhi), rtol=0, atol=1e-12)
        np.testing.assert_allclose(o_esacf.sacf(olo, ohi), np.array(xs), rtol=0, atol=1e-11)
    # 44.1 kHz default frame (N=2046) and a power-of-two frame (N=4096)
    x44 = poly_clip(11, sr=44100, length=3 * 4096)
    for n44 in (2046, 4096):
        fr = np.asarray(x44[:n44], dtype=np.float64)
        w = ref_wfir.wfir(fr, 44100, 12)
        hi = ref_lowpass.lowpass_filter(np.clip(ref_esacf._highpass_filter(w, 44100), 0, None), 44100, 1000)
        lo = ref_lowpass.lowpass_filter(w, 44100, 1000)
        s = ref_esacf._sacf([lo, hi])
        key = "fs44100_N%d" % n44
        stage[key + "/frames"] = fr[None]
        stage[key + "/wfir"] = w[None]
        stage[key + "/x_lo"] = lo[None]
        stage[key + "/x_hi"] = hi[None]
        stage[key + "/sacf"] = s[None]
    stage["provenance"] = np.array("ref-code")
    np.savez_compressed(os.path.join(HERE, "esacf_stages.npz"), **stage)

    # ------------------------------------------------------ G3 Harmonic Energy (ref-code)
    he = {"provenance": np.array("ref-code")}
    for name, x in clips.items():
        obj = chord_detection.MultipitchHarmonicEnergy(name)
        total = obj.compute_pitches()
        he[name + "/sum"] = np.array([total[i] for i in range(12)])
        he[name + "/repr"] = np.array(repr(total))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            he[name + "/key"] = np.array(total.key())
        F = int(np.ceil(len(x) / 8192))
        per = []
        for f in range(F):
            _CLIPS["__frame__"] = (x[f * 8192:(f + 1) * 8192], fs)
            c = chord_detection.MultipitchHarmonicEnergy("__frame__").compute_pitches()
            per.append([c[i] for i in range(12)])
        he[name + "/frames"] = np.array(per)
        np.testing.assert_allclose(o_he.he_frames(x, fs), np.array(per), rtol=1e-12, atol=0)
        np.testing.assert_allclose(o_he.he_compute(x, fs), he[name + "/sum"], rtol=1e-12, atol=0)
    # BASELINE shape: fs 44100, N 4096, hop 1024.  The reference only knows hop == N, so each
    # overlapped frame is pinned by running the reference on that frame alone.
    xT = poly_clip(21, sr=44100, length=7 * 1024 + 4096)
    per = []
    for f in range(8):
        _CLIPS["__frame__"] = (xT[f * 1024:f * 1024 + 4096], 44100)
        c = chord_detection.MultipitchHarmonicEnergy("__frame__", frame_size=4096).compute_pitches()
        per.append([c[i] for i in range(12)])
    he["T_fs44100_N4096_hop1024/x"] = xT
    he["T_fs44100_N4096_hop1024/frames"] = np.array(per)
    np.testing.assert_allclose(o_he.he_frames(xT, 44100, 4096, 1024), np.array(per), rtol=1e-12, atol=0)
    # non-default kwargs
    _CLIPS["__kw__"] = (clips["poly_seed2"], fs)
    c = chord_detection.MultipitchHarmonicEnergy("__kw__", frame_size=2048, num_harmonic=3, num_octave=3, num_bins=1).compute_pitches()
    he["kwargs_N2048_h3_o3_b1/sum"] = np.array([c[i] for i in range(12)])
    np.testing.assert_allclose(o_he.he_compute(clips["poly_seed2"], fs, 2048, num_harmonic=3, num_octave=3, num_bins=1),
                               he["kwargs_N2048_h3_o3_b1/sum"], rtol=1e-12)
    np.savez_compressed(os.path.join(HERE, "harmonic_energy.npz"), **he)

    # ------------------------------------------------------ ESACF end to end (ref-code+stub)
    es = {"provenance": np.array("ref-code+stub")}
    for mode, sfx in SPELLINGS:
        set_spelling(mode)
        for name, x in clips.items():
            obj = chord_detection.MultipitchESACF(name)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                total = obj.compute_pitches()
            es[name + "/sum" + sfx] = np.array([total[i] for i in range(12)])
            es[name + "/repr" + sfx] = np.array(repr(total))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                es[name + "/key" + sfx] = np.array(total.key())
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                mine = o_esacf.esacf_compute(x, fs, note_names=mode)
            np.testing.assert_allclose(mine, es[name + "/sum" + sfx], rtol=1e-6, atol=1e-9)
    set_spelling("unicode")
    # enhancement + peak stages on a few SACF frames
    sd = np.load(os.path.join(HERE, "esacf_stages.npz"))
    for name in ("piano_like_Cmaj", "poly_seed1"):
        enh, pk, pi = [], [], []
        for s in sd[name + "/sacf"]:
            e, _ = ref_esacf._esacf(s, 6, True)
            enh.append(e)
            p = sys.modules["peakutils"].indexes(e, thres=0.1, min_dist=10)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                q = sys.modules["peakutils"].interpolate(np.arange(e.shape[0]), e, ind=p)
            pk.append(np.pad(p, (0, 32 - len(p)), constant_values=-1))
            pi.append(np.pad(q, (0, 32 - len(q)), constant_values=np.nan))
        es[name + "/esacf"] = np.array(enh)
        es[name + "/peaks"] = np.array(pk)
        es[name + "/peaks_interp"] = np.array(pi)
    np.savez_compressed(os.path.join(HERE, "esacf_e2e.npz"), **es)

    # ------------------------------------------------------ Prime-multiF0 (ref-code: real matplotlib.mlab inside)
    from oracle import prime_multif0 as o_prime
    pr = {"provenance": np.array("ref-code+notes")}
    _CLIPS["__kw__"] = (clips["poly_seed1"], fs)
    for mode, sfx in SPELLINGS:
        set_spelling(mode)
        for name, x in clips.items():
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                total = chord_detection.MultipitchPrimeMultiF0(name).compute_pitches()
                pr[name + "/sum" + sfx] = np.array([total[i] for i in range(12)])
                pr[name + "/repr" + sfx] = np.array(repr(total))
                pr[name + "/key" + sfx] = np.array(total.key())
                np.testing.assert_allclose(o_prime.prime_compute(x, fs, note_names=mode), pr[name + "/sum" + sfx],
                                           rtol=1e-12, atol=0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c = chord_detection.MultipitchPrimeMultiF0("__kw__", num_harmonic=2, num_octave=3,
                                                       harmonic_multiples_elim=3, harmonic_elim_runs=3).compute_pitches()
        pr["kwargs_h2_o3_e3_r3/sum" + sfx] = np.array([c[i] for i in range(12)])
        np.testing.assert_allclose(o_prime.prime_compute(clips["poly_seed1"], fs, 2, 3, 3, 3, note_names=mode),
                                   pr["kwargs_h2_o3_e3_r3/sum" + sfx], rtol=1e-12)
    set_spelling("unicode")
    np.savez_compressed(os.path.join(HERE, "prime_multif0.npz"), **pr)

    # ------------------------------------------------------ Iterative F0 (ref-code: only closed-form librosa helpers)
    from oracle import iterative_f0 as o_if0
    it = {"provenance": np.array("ref-code+notes")}
    for mode, sfx in SPELLINGS:
        set_spelling(mode)
        for name in ("tone_Csharp3", "tone_E4", "tones_G3_Asharp4", "tones_G2_B2_Gsharp3", "piano_like_Cmaj",
                     "poly_seed1", "poly_seed2", "short_ragged"):
            x = clips[name]
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                obj = chord_detection.MultipitchIterativeF0(name)
                total = obj.compute_pitches()
                it[name + "/sum" + sfx] = np.array([total[i] for i in range(12)])
                it[name + "/repr" + sfx] = np.array(repr(total))
                it[name + "/key" + sfx] = np.array(total.key())
                per, Ut = o_if0.iterative_f0_frames(x, fs, note_names=mode)
                np.testing.assert_allclose(per.sum(0), it[name + "/sum" + sfx], rtol=1e-12, atol=0)
            # a thin slice of the summary spectrum of frame 0 (full rows are 16384 doubles each)
            it[name + "/ut0_head"] = Ut[0][:512].copy()
            it[name + "/frames" + sfx] = per
    set_spelling("unicode")
    np.savez_compressed(os.path.join(HERE, "iterative_f0.npz"), **it)
    for fn in ROUND5.values():
        fn(chord_detection, clips, fs)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
