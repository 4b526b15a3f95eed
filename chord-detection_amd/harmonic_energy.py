"""Method 2 -- drop-in for reference harmonic_energy.py:13-73, HIP-backed."""
from .chromagram import Chromagram
from .engine import get_engine
from .multipitch import Multipitch


class MultipitchHarmonicEnergy(Multipitch):
    def __init__(
        self, audio_path, frame_size=8192, num_harmonic=2, num_octave=2, num_bins=2, hop=None, fs=None, device=0,
        note_names="unicode",
    ):
        # note_names: accepted for a uniform constructor; this method indexes the chromagram by integer
        # (harmonic_energy.py:66-67) and never spells a note
        super().__init__(audio_path, fs=fs, device=device, note_names=note_names)
        self.frame_size = frame_size
        self.num_harmonic = num_harmonic
        self.num_octave = num_octave
        self.num_bins = num_bins
        self.hop = hop  # extension: overlapped frames (reference: hop == frame_size)
        # `dft_maxes` (harmonic_energy.py:36,65; its only reader is the reference's plot) is filled LAZILY: compute_pitches notes
        # that the list is due, and the first read of the attribute runs the debug tap of the C ABI (mpx_harmonic_energy_argmax:
        # a second copy of the signal and an untuned one-workgroup-per-frame launch) -- a caller that only wants the chromagram
        # never pays for it.  `record_dft_maxes = False` switches the tap off altogether.
        self.record_dft_maxes = True
        self._dft_maxes = []
        self._dft_maxes_due = False

    @staticmethod
    def display_name():
        return "Harmonic Energy (Stark, Plumbley)"

    @staticmethod
    def method_number():
        return 2

    def compute_pitches(self, display_plot_frame=-1):
        # display_plot_frame: accepted and ignored (matplotlib debugging aid in the reference)
        eng = get_engine(self.device)
        total = eng.harmonic_energy(
            self._samples(), self.fs, self.frame_size, self.hop, self.num_harmonic, self.num_octave, self.num_bins)
        self._dft_maxes = []                       # the reference appends per call; a new call starts a new list here
        self._dft_maxes_due = bool(self.record_dft_maxes)
        return Chromagram(total)

    @property
    def dft_maxes(self):
        """harmonic_energy.py:36,65: one (k0, best_ind, k1) per frame and window, in loop order, of the LAST compute_pitches call.
        Nothing the tap raises reaches the caller: the chromagram was returned already and does not depend on it."""
        if self._dft_maxes_due:
            self._dft_maxes_due = False
            import warnings
            if hasattr(self.x, "is_cuda"):
                warnings.warn("dft_maxes not recorded: the debug tap takes a host signal and this input is device-resident")
                return self._dft_maxes
            try:
                best, bounds = get_engine(self.device).harmonic_energy_argmax(
                    self.x, self.fs, self.frame_size, self.hop, self.num_harmonic, self.num_octave, self.num_bins)
            except Exception as e:   # a shape the (untuned) tap does not reach, no memory for the second copy, ...
                warnings.warn("dft_maxes not recorded: %s: %s" % (type(e).__name__, e))
                return self._dft_maxes
            none = -2 ** 31
            for row in best:
                self._dft_maxes.extend((int(k0), None if int(b) == none else int(b), int(k1))
                                       for b, (k0, k1) in zip(row, bounds))
        return self._dft_maxes

    @dft_maxes.setter
    def dft_maxes(self, value):
        self._dft_maxes, self._dft_maxes_due = value, False

    @classmethod
    def compute_batch(cls, clips, fs, frame_size=8192, num_harmonic=2, num_octave=2, num_bins=2, hop=None,
                      device=0):
        """Many clips in one launch -> list of Chromagram."""
        sums = get_engine(device).harmonic_energy_batch(clips, fs, frame_size, hop, num_harmonic, num_octave,
                                                        num_bins)
        return [Chromagram(s) for s in sums]
