"""GPU: the PCM_16 entry points of the C ABI (include/mpx.h, ABI 6: mpx_*_pcm16).  The reference's audio is 16-bit PCM that
librosa.load turns into int16 / 32768 as float32 (multipitch.py:24-30); these entry points take the int16 samples and scale
them on the device.  x / 32768 is exact in float32, so every method must return the SAME BITS as its float32 entry point fed
`pcm / 32768` -- asserted with array_equal, per frame and summed -- and a WAV file handed to the drop-in classes by path takes
this route and agrees with the oracle on the float32 samples."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = 22050


@pytest.fixture(scope="module")
def eng():
    import chord_detection_amd as cd
    return cd.get_engine(0)


def _pcm(seed, n):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / FS
    x = sum(a * np.sin(2 * np.pi * f * t) for a, f in ((0.4, 261.63), (0.3, 329.63), (0.2, 392.0))) + 0.01 * rng.standard_normal(n)
    pcm = np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)
    ends = np.array([-32768, 32767, 0, -1], dtype=np.int16)          # the ends of the int16 range convert exactly too
    pcm[:min(n, 4)] = ends[:min(n, 4)]
    return pcm


@pytest.mark.parametrize("n", [0, 1, 7, 8, 4097, 3 * 8192 + 5])
def test_every_method_is_bit_equal_to_its_float32_entry_point(eng, n):
    import chord_detection_amd as cd
    pcm = _pcm(n, n)
    x = pcm.astype(np.float32) / np.float32(32768.0)
    assert np.array_equal(cd.Pcm16(pcm).float32(), x) and np.array_equal(x.astype(np.float64) * 32768.0, pcm)   # exact
    p16 = cd.Pcm16(pcm)
    for frame, hop in ((8192, 8192), (4096, 1024), (1000, 1000)):
        a, af = eng.harmonic_energy(p16, FS, frame, hop, return_frames=True)
        b, bf = eng.harmonic_energy(x, FS, frame, hop, return_frames=True)
        assert np.array_equal(a, b) and np.array_equal(af, bf), (n, frame)
    a, af = eng.esacf(p16, FS, 1023, return_frames=True)
    b, bf = eng.esacf(x, FS, 1023, return_frames=True)
    assert np.array_equal(a, b) and np.array_equal(af, bf)
    assert np.array_equal(eng.prime_multif0(p16, FS), eng.prime_multif0(x, FS))
    if n:
        a, af = eng.iterative_f0(p16, FS, return_frames=True)
        b, bf = eng.iterative_f0(x, FS, return_frames=True)
        assert np.array_equal(a, b) and np.array_equal(af, bf)


def test_int16_samples_in_device_memory_and_plain_int16_arrays(eng):
    """The PCM entry points take device pointers like the float32 ones ("where the samples live"); a plain int16 ndarray is
    NOT PCM: it is cast to sample values like any other array (the scaling belongs to the file format, not the dtype)."""
    import ctypes as C
    import torch
    from chord_detection_amd import _lib
    pcm = _pcm(3, 20000)
    want = eng.harmonic_energy(pcm.astype(np.float32) / np.float32(32768.0), FS, 4096, 1024)
    t = torch.from_numpy(pcm).cuda()
    torch.cuda.synchronize()
    got = np.zeros(12)
    p = _lib.HeParams(2, 2, 2)
    rc = eng.lib.mpx_harmonic_energy_pcm16(eng.ctx, C.cast(C.c_void_p(t.data_ptr()), _lib._sp), t.numel(), FS, C.byref(p), 4096, 1024,
                                           None, got.ctypes.data_as(_lib._dp))
    assert rc == 0 and np.array_equal(got, want)
    as_values = eng.harmonic_energy(pcm, FS, 4096, 1024)                 # int16 ndarray: sample VALUES up to 32767
    assert np.array_equal(as_values, eng.harmonic_energy(pcm.astype(np.float32), FS, 4096, 1024)) and not np.array_equal(as_values, want)
    import chord_detection_amd as cd
    with pytest.raises(ValueError):
        cd.Pcm16(pcm.astype(np.int32))
    assert eng.lib.mpx_harmonic_energy_pcm16(eng.ctx, None, 5, FS, C.byref(p), 4096, 1024, None, got.ctypes.data_as(_lib._dp)) == _lib.MPX_EINVAL


def test_a_pcm16_wav_file_by_path_takes_the_int16_route_and_matches_the_oracle(eng, tmp_path):
    """multipitch.py:24-30 by path: a mono PCM_16 file at 22.05 kHz goes to the device as int16 (Multipitch._samples), the
    chromagram equals the one computed from the float32 samples bit for bit and the oracle's to the north star's 1e-5; a
    file at another rate (resampled on the host) and a stereo file keep the float32 route."""
    import chord_detection_amd as cd
    from chord_detection_amd import audio
    from oracle import harmonic_energy as o_he, chromagram as o_chroma
    pcm = _pcm(11, 5 * 8192 + 77)
    path = os.path.join(tmp_path, "clip.wav")
    audio.write_wav(path, pcm.astype(np.float64) / 32768.0, FS)
    back = audio.load_pcm16(path)
    assert back is not None and back[1] == FS and np.array_equal(back[0], pcm)          # the file holds exactly these samples
    obj = cd.MultipitchHarmonicEnergy(path)
    assert isinstance(obj._samples(), cd.Pcm16) and obj.fs == FS
    got = obj.compute_pitches()
    x = pcm.astype(np.float32) / np.float32(32768.0)
    assert np.array_equal(obj.x, x) and np.array_equal(obj.x, audio.load(path)[0])       # the reference's attribute, on demand
    assert np.array_equal(got.as_array(), cd.MultipitchHarmonicEnergy((x, FS)).compute_pitches().as_array())
    want = o_he.he_compute(x, FS)
    np.testing.assert_allclose(got.as_array(), want, rtol=1e-5)
    assert repr(got) == o_chroma.pack(want) and got.key() == o_chroma.detect_key(want)
    assert len(obj.dft_maxes) == 6 * 48                                                  # the lazy tap works on this route too
    for cls in (cd.MultipitchESACF, cd.MultipitchPrimeMultiF0, cd.MultipitchIterativeF0):
        assert np.array_equal(cls(path).compute_pitches().as_array(), cls((x, FS)).compute_pitches().as_array()), cls
    # another rate: resampled on the host -> float32 route
    path2 = os.path.join(tmp_path, "clip44.wav")
    audio.write_wav(path2, pcm.astype(np.float64) / 32768.0, 44100)
    assert audio.load_pcm16(path2) is None and audio.load_pcm16(path2, sr=None)[1] == 44100
    o2 = cd.MultipitchHarmonicEnergy(path2)
    assert not isinstance(o2._samples(), cd.Pcm16) and o2.fs == FS and o2.x.dtype == np.float32
    # samples assigned by the caller replace the file's
    obj.x = x[:8192]
    assert not isinstance(obj._samples(), cd.Pcm16) and obj._samples().shape[0] == 8192
