"""Restatement of the third-party routines the reference calls on the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

None of these live under /root/reference and none is installed in the build
image (librosa, peakutils: ``ModuleNotFoundError``; requirements.txt:3,6 leave
them unpinned).  Each function restates the library's *published* algorithm
(librosa 0.10.x, peakutils 1.3.x, MINPACK lmdif as wrapped by
scipy.optimize.curve_fit) and names the reference call site that needs it.

  hz_to_note / note_to_hz / cqt_frequencies / tone : closed forms, unambiguous.
  time_stretch   : reference esacf.py:121            -> PARITY UNPINNED
  peak_indexes   : reference esacf.py:56             -> PARITY UNPINNED
  peak_interpolate / gaussian_fit : esacf.py:60      -> PARITY UNPINNED
                   (cross-checked in tests against the real scipy curve_fit)
"""
import math

import numpy as np

NOTE_NAMES_UNICODE = ["C", "C♯", "D", "D♯", "E", "F", "F♯", "G", "G♯", "A", "A♯", "B"]  # librosa >= 0.8
NOTE_NAMES_ASCII = ["C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"]    # librosa < 0.8


# --------------------------------------------------------------------------
# librosa closed forms (call sites: esacf.py:68, harmonic_energy.py:33,
# periodicity.py:107, prime_multif0.py:45,70, tests/gen_test_clips.py:14-41)
# --------------------------------------------------------------------------
def hz_to_midi(f):
    return 12.0 * (np.log2(np.asanyarray(f, dtype=np.float64)) - np.log2(440.0)) + 69.0


def hz_to_pitch_class(f):
    """int(round(midi)) % 12; raises like librosa.hz_to_note: NaN -> ValueError,
    inf -> OverflowError (python int() semantics)."""
    midi = float(hz_to_midi(f))
    return int(np.round(midi)) % 12


def hz_to_note(f, octave=False, unicode=True):
    """librosa.hz_to_note(f, octave=...).  The spelling of sharps changed in librosa 0.8 ("C#" -> "C♯");
    `unicode=False` is the older one (see oracle/esacf.py on what that does to the reference's chromagram)."""
    pc = hz_to_pitch_class(f)
    name = (NOTE_NAMES_UNICODE if unicode else NOTE_NAMES_ASCII)[pc]
    if octave:
        name += str(int(np.round(float(hz_to_midi(f)))) // 12 - 1)
    return name


def note_to_hz(note):
    pitch = {"C": 0, "D": 2, "E": 4, "F": 5, "G": 7, "A": 9, "B": 11}[note[0].upper()]
    rest = note[1:]
    while rest and rest[0] in "#♯b♭":
        pitch += 1 if rest[0] in "#♯" else -1
        rest = rest[1:]
    octave = int(rest) if rest else 0
    midi = 12 * (octave + 1) + pitch
    return 440.0 * (2.0 ** ((midi - 69.0) / 12.0))


def cqt_frequencies(n_bins, fmin, bins_per_octave=12):
    return fmin * 2.0 ** (np.arange(n_bins, dtype=np.float64) / bins_per_octave)


def tone(frequency, sr=22050, length=None):
    """librosa.tone: cos(2 pi f n / sr + phi), phi = -pi/2 by default."""
    return np.cos(2.0 * np.pi * frequency * np.arange(int(length)) / sr - 0.5 * np.pi)


# --------------------------------------------------------------------------
# librosa.effects.time_stretch  (reference esacf.py:121)  -- UNPINNED
# stft(n_fft=2048, hop=512, periodic hann, center, zero pad) -> phase_vocoder
# -> istft(length=round(len/rate)).
# --------------------------------------------------------------------------
N_FFT = 2048
HOP = 512


def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft(y, n_fft=N_FFT, hop=HOP):
    y = np.asarray(y, dtype=np.float64)
    ypad = np.concatenate([np.zeros(n_fft // 2), y, np.zeros(n_fft // 2)])
    n_frames = 1 + (ypad.shape[0] - n_fft) // hop
    w = hann_periodic(n_fft)
    D = np.empty((n_fft // 2 + 1, n_frames), dtype=np.complex128)
    for t in range(n_frames):
        D[:, t] = np.fft.rfft(w * ypad[t * hop:t * hop + n_fft])
    return D


def phase_vocoder(D, rate, hop=HOP):
    n_bins, n_frames = D.shape
    time_steps = np.arange(0, n_frames, rate, dtype=np.float64)
    out = np.zeros((n_bins, len(time_steps)), dtype=D.dtype)
    phi_advance = np.linspace(0, np.pi * hop, n_bins)
    phase_acc = np.angle(D[:, 0])
    Dp = np.concatenate([D, np.zeros((n_bins, 2), dtype=D.dtype)], axis=1)
    for t, step in enumerate(time_steps):
        c0 = Dp[:, int(step)]
        c1 = Dp[:, int(step) + 1]
        alpha = np.mod(step, 1.0)
        mag = (1.0 - alpha) * np.abs(c0) + alpha * np.abs(c1)
        out[:, t] = mag * (np.cos(phase_acc) + 1j * np.sin(phase_acc))
        dphase = np.angle(c1) - np.angle(c0) - phi_advance
        dphase = dphase - 2.0 * np.pi * np.round(dphase / (2.0 * np.pi))
        phase_acc = phase_acc + phi_advance + dphase
    return out


def istft(D, length, n_fft=N_FFT, hop=HOP):
    w = hann_periodic(n_fft)
    n_frames = D.shape[1]
    padded_length = length + 2 * (n_fft // 2)
    n_frames = min(n_frames, int(np.ceil(padded_length / hop)))
    total = n_fft + hop * (n_frames - 1)
    y = np.zeros(total)
    wss = np.zeros(total)
    for t in range(n_frames):
        y[t * hop:t * hop + n_fft] += w * np.fft.irfft(D[:, t], n=n_fft)
        wss[t * hop:t * hop + n_fft] += w * w
    y = y[n_fft // 2:]
    wss = wss[n_fft // 2:]
    out = np.zeros(length)
    m = min(length, y.shape[0])
    yy = y[:m].copy()
    ww = wss[:m]
    nz = ww > np.finfo(np.float64).tiny
    yy[nz] /= ww[nz]
    out[:m] = yy
    return out


def time_stretch(y, rate):
    y = np.asarray(y, dtype=np.float64)
    D = stft(y)
    Ds = phase_vocoder(D, rate)
    return istft(Ds, int(round(y.shape[-1] / rate)))


def time_stretch_is_truncation(n):
    """True when the STFT of an n-sample signal has <= 2 frames, in which case
    time_stretch(y, r>=2) == y[:round(n/r)] (single-frame ISTFT undoes its own
    window)."""
    return 1 + n // HOP <= 2


# --------------------------------------------------------------------------
# peakutils.indexes  (reference esacf.py:56)  -- UNPINNED
# --------------------------------------------------------------------------
def peak_indexes(y, thres=0.3, min_dist=1):
    y = np.asarray(y, dtype=np.float64)
    thres = thres * (np.max(y) - np.min(y)) + np.min(y)
    min_dist = int(min_dist)
    dy = np.diff(y)
    zeros, = np.where(dy == 0)
    if len(zeros) == len(y) - 1:
        return np.array([], dtype=np.int64)
    if len(zeros):
        zeros_diff = np.diff(zeros)
        zeros_diff_not_one, = np.add(np.where(zeros_diff != 1), 1)
        zero_plateaus = np.split(zeros, zeros_diff_not_one)
        if zero_plateaus[0][0] == 0:
            dy[zero_plateaus[0]] = dy[zero_plateaus[0][-1] + 1]
            zero_plateaus.pop(0)
        if len(zero_plateaus) and zero_plateaus[-1][-1] == len(dy) - 1:
            dy[zero_plateaus[-1]] = dy[zero_plateaus[-1][0] - 1]
            zero_plateaus.pop(-1)
        for plateau in zero_plateaus:
            median = np.median(plateau)
            dy[plateau[plateau < median]] = dy[plateau[0] - 1]
            dy[plateau[plateau >= median]] = dy[plateau[-1] + 1]
    peaks = np.where((np.hstack([dy, 0.0]) < 0.0)
                     & (np.hstack([0.0, dy]) > 0.0)
                     & (np.greater(y, thres)))[0]
    if peaks.size > 1 and min_dist > 1:
        highest = peaks[np.argsort(y[peaks])][::-1]
        rem = np.ones(y.size, dtype=bool)
        rem[peaks] = False
        for peak in highest:
            if not rem[peak]:
                sl = slice(max(0, peak - min_dist), peak + min_dist + 1)
                rem[sl] = True
                rem[peak] = False
        peaks = np.arange(y.size)[~rem]
    return peaks.astype(np.int64)


# --------------------------------------------------------------------------
# peakutils.interpolate -> gaussian_fit -> scipy curve_fit -> MINPACK lmdif
# (reference esacf.py:60)  -- UNPINNED; restated from MINPACK's published
# algorithm (Moré 1978: lmdif / lmpar / qrfac / qrsolv / fdjac2).
# --------------------------------------------------------------------------
_EPSMCH = float(np.finfo(np.float64).eps)
_DWARF = float(np.finfo(np.float64).tiny)
GAUSS_EPS = _EPSMCH  # peakutils: eps = np.finfo(float).eps


def gaussian(x, ampl, center, dev):
    return ampl * np.exp(-((x - float(center)) ** 2) / (2.0 * dev ** 2 + GAUSS_EPS))


def _enorm(v):
    return math.sqrt(float(np.dot(v, v)))


def _qrfac(a):
    """Householder QR with column pivoting (MINPACK qrfac). a is modified."""
    m, n = a.shape
    acnorm = np.array([_enorm(a[:, j]) for j in range(n)])
    rdiag = acnorm.copy()
    wa = acnorm.copy()
    ipvt = list(range(n))
    for j in range(min(m, n)):
        kmax = j + int(np.argmax(rdiag[j:]))
        if kmax != j:
            a[:, [j, kmax]] = a[:, [kmax, j]]
            rdiag[kmax] = rdiag[j]
            wa[kmax] = wa[j]
            ipvt[j], ipvt[kmax] = ipvt[kmax], ipvt[j]
        ajnorm = _enorm(a[j:, j])
        if ajnorm != 0.0:
            if a[j, j] < 0.0:
                ajnorm = -ajnorm
            a[j:, j] /= ajnorm
            a[j, j] += 1.0
            for k in range(j + 1, n):
                s = float(np.dot(a[j:, j], a[j:, k]))
                temp = s / a[j, j]
                a[j:, k] -= temp * a[j:, j]
                if rdiag[k] != 0.0:
                    temp = a[j, k] / rdiag[k]
                    rdiag[k] *= math.sqrt(max(0.0, 1.0 - temp * temp))
                    if 0.05 * (rdiag[k] / wa[k]) ** 2 <= _EPSMCH:
                        rdiag[k] = _enorm(a[j + 1:, k])
                        wa[k] = rdiag[k]
        rdiag[j] = -ajnorm
    return ipvt, rdiag, acnorm


def _qrsolv(n, r, ipvt, diag, qtb):
    x = np.zeros(n)
    sdiag = np.zeros(n)
    wa = np.zeros(n)
    for j in range(n):
        for i in range(j, n):
            r[i, j] = r[j, i]
        x[j] = r[j, j]
        wa[j] = qtb[j]
    for j in range(n):
        l = ipvt[j]
        if diag[l] != 0.0:
            sdiag[j:] = 0.0
            sdiag[j] = diag[l]
            qtbpj = 0.0
            for k in range(j, n):
                if sdiag[k] == 0.0:
                    continue
                if abs(r[k, k]) < abs(sdiag[k]):
                    cotan = r[k, k] / sdiag[k]
                    sin = 0.5 / math.sqrt(0.25 + 0.25 * cotan * cotan)
                    cos = sin * cotan
                else:
                    tan = sdiag[k] / r[k, k]
                    cos = 0.5 / math.sqrt(0.25 + 0.25 * tan * tan)
                    sin = cos * tan
                r[k, k] = cos * r[k, k] + sin * sdiag[k]
                temp = cos * wa[k] + sin * qtbpj
                qtbpj = -sin * wa[k] + cos * qtbpj
                wa[k] = temp
                for i in range(k + 1, n):
                    temp = cos * r[i, k] + sin * sdiag[i]
                    sdiag[i] = -sin * r[i, k] + cos * sdiag[i]
                    r[i, k] = temp
        sdiag[j] = r[j, j]
        r[j, j] = x[j]
    nsing = n
    for j in range(n):
        if sdiag[j] == 0.0 and nsing == n:
            nsing = j
        if nsing < n:
            wa[j] = 0.0
    for k in range(nsing):
        j = nsing - 1 - k
        s = 0.0
        for i in range(j + 1, nsing):
            s += r[i, j] * wa[i]
        wa[j] = (wa[j] - s) / sdiag[j]
    for j in range(n):
        x[ipvt[j]] = wa[j]
    return x, sdiag


def _lmpar(n, r, ipvt, diag, qtb, delta, par):
    x = np.zeros(n)
    sdiag = np.zeros(n)
    nsing = n
    wa1 = np.array(qtb[:n], dtype=np.float64)
    for j in range(n):
        if r[j, j] == 0.0 and nsing == n:
            nsing = j
        if nsing < n:
            wa1[j] = 0.0
    for k in range(nsing):
        j = nsing - 1 - k
        wa1[j] /= r[j, j]
        temp = wa1[j]
        for i in range(j):
            wa1[i] -= r[i, j] * temp
    for j in range(n):
        x[ipvt[j]] = wa1[j]
    it = 0
    wa2 = diag * x
    dxnorm = _enorm(wa2)
    fp = dxnorm - delta
    if fp <= 0.1 * delta:
        return 0.0, x, sdiag
    parl = 0.0
    if nsing >= n:
        for j in range(n):
            l = ipvt[j]
            wa1[j] = diag[l] * (wa2[l] / dxnorm)
        for j in range(n):
            s = 0.0
            for i in range(j):
                s += r[i, j] * wa1[i]
            wa1[j] = (wa1[j] - s) / r[j, j]
        temp = _enorm(wa1)
        parl = ((fp / delta) / temp) / temp
    for j in range(n):
        s = 0.0
        for i in range(j + 1):
            s += r[i, j] * qtb[i]
        l = ipvt[j]
        wa1[j] = s / diag[l]
    gnorm = _enorm(wa1)
    paru = gnorm / delta
    if paru == 0.0:
        paru = _DWARF / min(delta, 0.1)
    par = max(par, parl)
    par = min(par, paru)
    if par == 0.0:
        par = gnorm / dxnorm
    while True:
        it += 1
        if par == 0.0:
            par = max(_DWARF, 0.001 * paru)
        temp = math.sqrt(par)
        wa1 = temp * diag
        x, sdiag = _qrsolv(n, r, ipvt, wa1, qtb)
        wa2 = diag * x
        dxnorm = _enorm(wa2)
        temp = fp
        fp = dxnorm - delta
        if abs(fp) <= 0.1 * delta or (parl == 0.0 and fp <= temp and temp < 0.0) or it == 10:
            break
        for j in range(n):
            l = ipvt[j]
            wa1[j] = diag[l] * (wa2[l] / dxnorm)
        for j in range(n):
            wa1[j] /= sdiag[j]
            temp = wa1[j]
            for i in range(j + 1, n):
                wa1[i] -= r[i, j] * temp
        temp = _enorm(wa1)
        parc = ((fp / delta) / temp) / temp
        if fp > 0.0:
            parl = max(parl, par)
        if fp < 0.0:
            paru = min(paru, par)
        par = max(parl, par + parc)
    return par, x, sdiag


def lmdif(fcn, x0, ftol=1.49012e-8, xtol=1.49012e-8, gtol=0.0, maxfev=None,
          epsfcn=None, factor=100.0):
    """MINPACK lmdif as driven by scipy.optimize.leastsq/curve_fit defaults.
    Returns (x, info, nfev); info in 1..4 means converged."""
    x = np.array(x0, dtype=np.float64)
    n = x.shape[0]
    if maxfev is None:
        maxfev = 200 * (n + 1)
    if epsfcn is None:
        epsfcn = _EPSMCH
    fvec = np.asarray(fcn(x), dtype=np.float64)
    m = fvec.shape[0]
    nfev = 1
    fnorm = _enorm(fvec)
    par = 0.0
    it = 1
    info = 0
    eps = math.sqrt(max(epsfcn, _EPSMCH))
    diag = np.ones(n)
    delta = 0.0
    xnorm = 0.0
    while True:
        # forward-difference jacobian (fdjac2)
        fjac = np.empty((m, n))
        for j in range(n):
            temp = x[j]
            h = eps * abs(temp)
            if h == 0.0:
                h = eps
            x[j] = temp + h
            wa = np.asarray(fcn(x), dtype=np.float64)
            x[j] = temp
            fjac[:, j] = (wa - fvec) / h
        nfev += n
        ipvt, wa1, wa2 = _qrfac(fjac)
        if it == 1:
            diag = np.where(wa2 != 0.0, wa2, 1.0)
            wa3 = diag * x
            xnorm = _enorm(wa3)
            delta = factor * xnorm
            if delta == 0.0:
                delta = factor
        wa4 = fvec.copy()
        qtf = np.zeros(n)
        for j in range(n):
            if fjac[j, j] != 0.0:
                s = float(np.dot(fjac[j:, j], wa4[j:]))
                temp = -s / fjac[j, j]
                wa4[j:] += fjac[j:, j] * temp
            fjac[j, j] = wa1[j]
            qtf[j] = wa4[j]
        gnorm = 0.0
        if fnorm != 0.0:
            for j in range(n):
                l = ipvt[j]
                if wa2[l] != 0.0:
                    s = 0.0
                    for i in range(j + 1):
                        s += fjac[i, j] * (qtf[i] / fnorm)
                    gnorm = max(gnorm, abs(s / wa2[l]))
        if gnorm <= gtol:
            info = 4
            break
        diag = np.maximum(diag, wa2)
        r = fjac[:n, :n].copy()
        while True:
            rr = r.copy()
            par, p, _sd = _lmpar(n, rr, ipvt, diag, qtf, delta, par)
            p = -p
            xnew = x + p
            wa3 = diag * p
            pnorm = _enorm(wa3)
            if it == 1:
                delta = min(delta, pnorm)
            fnew = np.asarray(fcn(xnew), dtype=np.float64)
            nfev += 1
            fnorm1 = _enorm(fnew)
            actred = -1.0
            if 0.1 * fnorm1 < fnorm:
                actred = 1.0 - (fnorm1 / fnorm) ** 2
            wa3 = np.zeros(n)
            for j in range(n):
                l = ipvt[j]
                temp = p[l]
                for i in range(j + 1):
                    wa3[i] += r[i, j] * temp
            temp1 = _enorm(wa3) / fnorm
            temp2 = (math.sqrt(par) * pnorm) / fnorm
            prered = temp1 * temp1 + temp2 * temp2 / 0.5
            dirder = -(temp1 * temp1 + temp2 * temp2)
            ratio = actred / prered if prered != 0.0 else 0.0
            if ratio <= 0.25:
                if actred >= 0.0:
                    temp = 0.5
                else:
                    temp = 0.5 * dirder / (dirder + 0.5 * actred)
                if 0.1 * fnorm1 >= fnorm or temp < 0.1:
                    temp = 0.1
                delta = temp * min(delta, pnorm / 0.1)
                par = par / temp
            elif par == 0.0 or ratio >= 0.75:
                delta = pnorm / 0.5
                par = 0.5 * par
            if ratio >= 1e-4:
                x = xnew
                fvec = fnew
                xnorm = _enorm(diag * x)
                fnorm = fnorm1
                it += 1
            if abs(actred) <= ftol and prered <= ftol and 0.5 * ratio <= 1.0:
                info = 1
            if delta <= xtol * xnorm:
                info = 2
            if (abs(actred) <= ftol and prered <= ftol and 0.5 * ratio <= 1.0
                    and info == 2):
                info = 3
            if info != 0:
                break
            if nfev >= maxfev:
                info = 5
            if abs(actred) <= _EPSMCH and prered <= _EPSMCH and 0.5 * ratio <= 1.0:
                info = 6
            if delta <= _EPSMCH * xnorm:
                info = 7
            if gnorm <= _EPSMCH:
                info = 8
            if info != 0:
                break
            if ratio >= 1e-4:
                break
        if info != 0:
            break
    return x, info, nfev


def gaussian_fit_center(xs, ys):
    """peakutils.gaussian_fit(center_only=True). Raises RuntimeError like
    peakutils/curve_fit when the fit cannot be done."""
    xs = np.asarray(xs, dtype=np.float64)
    ys = np.asarray(ys, dtype=np.float64)
    if len(xs) < 3:
        raise RuntimeError("At least 3 points required for Gaussian fitting")
    p0 = [float(np.max(ys)), float(xs[0]), float((xs[1] - xs[0]) * 5)]

    def resid(p):
        with np.errstate(all="ignore"):
            return gaussian(xs, p[0], p[1], p[2]) - ys

    p, info, _ = lmdif(resid, p0)
    if info not in (1, 2, 3, 4):
        raise RuntimeError("Optimal parameters not found")
    return float(p[1])


def peak_interpolate(x, y, ind, width=10):
    out = []
    for i in ind:
        i = int(i)
        sl = slice(i - width, i + width + 1)
        try:
            out.append(gaussian_fit_center(x[sl], y[sl]))
        except Exception:
            pass
    return np.array(out)
