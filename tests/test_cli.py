"""CLI: flag parsing on CPU; an end-to-end run over a generated WAV on the GPU."""
import numpy as np
import pytest


def test_cli_rejects_unknown_method(tmp_path):
    from chord_detection_amd import chord_detect
    with pytest.raises(ValueError):
        chord_detect.main_cli(["--method", "9", str(tmp_path / "nope.wav")])


def test_wav_loader_roundtrip(tmp_path):
    from chord_detection_amd import audio
    t = np.arange(22050) / 22050.0
    x = 0.5 * np.sin(2 * np.pi * 440 * t)
    audio.write_wav(tmp_path / "a.wav", x, 22050)
    y, fs = audio.load(tmp_path / "a.wav")
    assert fs == 22050 and y.dtype == np.float32 and y.shape == (22050,)
    assert np.max(np.abs(y - x)) < 1.0 / 32768 + 1e-7
    audio.write_wav(tmp_path / "b.wav", np.sin(2 * np.pi * 440 * np.arange(44100) / 44100.0), 44100)
    z, fs2 = audio.load(tmp_path / "b.wav")        # resampled to 22050 like librosa.load's default
    assert fs2 == 22050 and abs(z.shape[0] - 22050) <= 1
    # 48 kHz -> 22.05 kHz (up 147 / down 320): polyphase, milliseconds, and the tone survives
    import time
    t48 = np.arange(2 * 48000) / 48000.0
    audio.write_wav(tmp_path / "c.wav", 0.5 * np.sin(2 * np.pi * 440 * t48), 48000)
    t0 = time.perf_counter()
    u, fs3 = audio.load(tmp_path / "c.wav")
    assert time.perf_counter() - t0 < 5.0 and fs3 == 22050 and u.shape == (44100,)
    tt = np.arange(44100) / 22050.0
    assert np.max(np.abs(u[2000:-2000] - 0.5 * np.sin(2 * np.pi * 440 * tt[2000:-2000]))) < 1e-3


@pytest.mark.gpu
def test_cli_all_methods_on_generated_clip(tmp_path, capsys):
    from chord_detection_amd import audio, chord_detect
    t = np.arange(44100) / 22050.0
    x = 0.3 * (np.sin(2 * np.pi * 261.63 * t) + np.sin(2 * np.pi * 329.63 * t) + np.sin(2 * np.pi * 392.0 * t))
    path = tmp_path / "cmaj.wav"
    audio.write_wav(path, x, 22050)
    assert chord_detect.main_cli(["--method", "-1", "--key", str(path)]) == 0
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 12
    assert out[0] == "1 - ESACF (Tolonen, Karjalainen)" and out[3] == "2 - Harmonic Energy (Stark, Plumbley)"
    assert out[6] == "3 - Iterative F0 (Klapuri, Anssi)" and out[9] == "4 - Prime-multiF0 (Camacho, Kaver-Oreamuno)"
    for i in (1, 4, 7, 10):
        assert len(out[i]) == 12 and out[i].isdigit()
    # ... and the printed strings / keys are the oracle's for the samples the loader hands over, in both note spellings
    import warnings
    from oracle import chromagram as o_chroma, esacf as o_esacf, harmonic_energy as o_he
    from oracle import iterative_f0 as o_if0, prime_multif0 as o_prime
    y, fs = audio.load(path)
    for mode in ("unicode", "ascii"):
        assert chord_detect.main_cli(["--method", "-1", "--key", "--note-names", mode, str(path)]) == 0
        got = capsys.readouterr().out.strip().splitlines()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = [o_esacf.esacf_compute(y, fs, note_names=mode), o_he.he_compute(y, fs),
                    o_if0.iterative_f0_compute(y, fs, note_names=mode), o_prime.prime_compute(y, fs, note_names=mode)]
            for m, w in enumerate(want):
                assert got[3 * m + 1] == o_chroma.pack(w), (mode, m + 1, got[3 * m + 1], o_chroma.pack(w))
                assert got[3 * m + 2] == o_chroma.detect_key(w), (mode, m + 1)


def test_cli_note_names_flag_parses(tmp_path):
    from chord_detection_amd import chord_detect
    with pytest.raises(SystemExit):
        chord_detect.main_cli(["--note-names", "latin1", str(tmp_path / "nope.wav")])
