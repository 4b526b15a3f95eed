"""Method 3 -- drop-in for reference iterative_f0.py:20-96 (+ periodicity.py), HIP-backed."""
import math

from .chromagram import Chromagram
from .engine import get_engine
from .multipitch import Multipitch


class IterativeF0PeriodicityAnalysis:
    """periodicity.py:8-47: the period-search object the reference keeps in MultipitchIterativeF0.periodicity_estimator.
    `compute(Uk)` runs the HIP period search on ONE summary spectrum of 2 x window_size bins
    (mpx_iterative_f0_periodicity_voices) and returns what periodicity.py:112 returns: the Chromagram and the pair
    (voicesaliences, voiceperiods) -- two arrays of `max_voices` entries, the detected voices' saliences and periods in
    seconds, zeros for the voices the search did not find -- also kept as attributes of those names, like the reference."""

    def __init__(self, fs, window_size, max_voices=4, tau_min=1.0 / 2100.0, tau_max=1.0 / 40.0, tau_prec=0.0000001, Q=20,
                 M=20, epsilon1=20, epsilon2=320, gamma=0.66, device=0, note_names="unicode"):
        self.fs = fs
        self.window_size = window_size
        self.K = window_size / fs
        self.max_voices, self.tau_min, self.tau_max, self.tau_prec = max_voices, tau_min, tau_max, tau_prec
        self.Q, self.M, self.epsilon1, self.epsilon2, self.gamma = Q, M, epsilon1, epsilon2, gamma
        self.device, self.note_names = device, note_names

    def compute(self, Uk):
        import numpy
        Uk = numpy.asarray(Uk, dtype=numpy.float64)
        if Uk.ndim != 1:   # periodicity.py:48 takes ONE summary spectrum; the engine call underneath takes rows of them
            raise ValueError("compute takes one summary spectrum of 2 x window_size bins (got shape %s)" % (Uk.shape,))
        rows, sal, per = get_engine(self.device).iterative_f0_periodicity(
            Uk, self.fs, return_voices=True, frame_size=self.window_size, max_voices=self.max_voices, tau_min=self.tau_min,
            tau_max=self.tau_max, tau_prec=self.tau_prec, Q=self.Q, M=self.M, epsilon1=self.epsilon1, epsilon2=self.epsilon2,
            gamma=self.gamma, note_names=self.note_names)
        self.voicesaliences, self.voiceperiods = sal[0], per[0]                      # periodicity.py:42-43
        return Chromagram(rows[0]), (self.voicesaliences.copy(), self.voiceperiods.copy())   # periodicity.py:112


class MultipitchIterativeF0(Multipitch):
    def __init__(
        self,
        audio_path,
        frame_size=8192,
        power=1.0,
        channels=70,
        zeta0=2.3,
        zeta1=0.39,
        peak_thresh=0.5,
        peak_min_dist=10,
        harmonic_multiples_elim=5,
        fs=None,
        device=0,
        note_names="unicode",
    ):
        super().__init__(audio_path, fs=fs, device=device, note_names=note_names)
        self.frame_size = frame_size
        self.num_frames = math.ceil(self._samples().shape[0] / self.frame_size)
        self.power = power
        self.num_channels = channels
        self.zeta0, self.zeta1 = zeta0, zeta1
        self.channels = [
            229 * (10 ** ((zeta1 * c + zeta0) / 21.4) - 1) for c in range(channels)
        ]
        # accepted for signature parity; the reference never reads them (iterative_f0.py:40-42)
        self.peak_thresh = peak_thresh
        self.peak_min_dist = peak_min_dist
        self.harmonic_multiples_elim = harmonic_multiples_elim
        self.periodicity_estimator = IterativeF0PeriodicityAnalysis(self.fs, self.frame_size, device=device,
                                                                    note_names=note_names)   # iterative_f0.py:44

    @staticmethod
    def display_name():
        return "Iterative F0 (Klapuri, Anssi)"

    @staticmethod
    def method_number():
        return 3

    def compute_pitches(self, display_plot_frame=-1):
        # hours of audio go through the same call: above the context's workspace cap (include/mpx.h
        # MPX_OPT_IF0_WORKSPACE_BYTES, 32 GiB of front-end output = ~21 min at 44.1 kHz) the library advances its chunks in
        # time slices and carries the filter state over, so nothing has to be cut here
        total = get_engine(self.device).iterative_f0(
            self._samples(), self.fs, frame_size=self.frame_size, power=self.power, channels=self.num_channels,
            zeta0=self.zeta0, zeta1=self.zeta1, note_names=self.note_names)
        return Chromagram(total)

    @classmethod
    def compute_batch(cls, clips, fs, frame_size=8192, power=1.0, channels=70, zeta0=2.3, zeta1=0.39, device=0,
                      note_names="unicode"):
        sums = get_engine(device).iterative_f0_batch(clips, fs, frame_size=frame_size, power=power,
                                                     channels=channels, zeta0=zeta0, zeta1=zeta1,
                                                     note_names=note_names)
        return [Chromagram(s) for s in sums]
