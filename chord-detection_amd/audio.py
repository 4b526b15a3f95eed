"""Minimal audio ingest for the drop-in API: PCM WAV -> mono float32.

The reference calls librosa.load(path) (multipitch.py:25), i.e. decode +
resample to 22050 Hz mono float32.  Decoding/resampling is I/O, not part of the
data-parallel path, and librosa's soxr resampler cannot be matched bit for bit;
this loader decodes PCM and, when the file is not already at the requested rate,
resamples with a windowed-sinc polyphase filter.  Callers that need parity with
the reference pass (x, fs) arrays instead of paths.
"""
import math
import wave

import numpy as np


def _resample(x, sr_in, sr_out):
    """Windowed-sinc polyphase resampler: scipy.signal.resample_poly evaluates only the outputs that are kept (one
    sub-filter per output phase), so 48 kHz -> 22.05 kHz (up 147, down 320, 10 241 taps) costs 32 multiply-adds per
    output sample instead of a 10 241-tap convolution over the zero-stuffed signal."""
    if sr_in == sr_out:
        return x
    import scipy.signal
    g = math.gcd(int(sr_in), int(sr_out))
    up, down = int(sr_out) // g, int(sr_in) // g
    cutoff = 1.0 / max(up, down)
    half = 16 * max(up, down)
    n = np.arange(-half, half + 1)
    h = cutoff * np.sinc(cutoff * n) * np.hanning(2 * half + 1)
    h = h / h.sum()          # unit DC gain; resample_poly applies the factor `up` itself
    return scipy.signal.resample_poly(np.asarray(x, dtype=np.float64), up, down, window=h).astype(np.float32)


def load_pcm16(path, sr=22050):
    """(int16 samples, rate) when the file is mono PCM_16 already at `sr` (or `sr` is None) -- the samples exactly as the file
    holds them, for the library's PCM_16 entry points (include/mpx.h: x / 32768 on the device, half the bytes over PCIe) -- else
    None: anything that has to be down-mixed or resampled goes through load()."""
    with wave.open(str(path), "rb") as w:
        nch, width, rate, nframes = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        if nch != 1 or width != 2 or (sr is not None and rate != sr):
            return None
        raw = w.readframes(nframes)
    return np.frombuffer(raw, dtype="<i2").astype(np.int16, copy=False), int(rate)


def load(path, sr=22050):
    """Returns (float32 mono samples, sample rate) like librosa.load(path)."""
    with wave.open(str(path), "rb") as w:
        nch, width, rate, nframes = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(nframes)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        x = v.astype(np.float32) / 8388608.0
    else:
        raise ValueError("unsupported WAV sample width %d" % width)
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1).astype(np.float32)
    if sr is not None and rate != sr:
        x = _resample(x.astype(np.float64), rate, sr)
        rate = sr
    return np.ascontiguousarray(x, dtype=np.float32), int(rate)


def write_wav(path, x, sr):
    """PCM_16 writer (for the test-clip generator)."""
    x = np.clip(np.asarray(x, dtype=np.float64), -1.0, 32767.0 / 32768.0)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sr))
        w.writeframes((x * 32768.0).astype("<i2").tobytes())
