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
        self.record_dft_maxes = True   # fill `dft_maxes` in compute_pitches like the reference does (a second, untuned launch)

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
            self.x, self.fs, self.frame_size, self.hop, self.num_harmonic, self.num_octave, self.num_bins)
        # harmonic_energy.py:36,65: one (k0, best_ind, k1) per frame and window, in loop order; its only reader is the
        # reference's plot.  Filled from the debug tap of the C ABI (mpx_harmonic_energy_argmax); switch it off with
        # `record_dft_maxes = False` where only the chromagram is wanted.
        self.dft_maxes = []
        if self.record_dft_maxes and not hasattr(self.x, "is_cuda"):
            try:
                best, bounds = eng.harmonic_energy_argmax(self.x, self.fs, self.frame_size, self.hop, self.num_harmonic,
                                                          self.num_octave, self.num_bins)
            except NotImplementedError as e:   # a shape the (untuned) tap does not reach: the chromagram does not depend on it
                import warnings
                warnings.warn("dft_maxes not recorded: %s" % e)
                return Chromagram(total)
            none = -2 ** 31
            for row in best:
                self.dft_maxes.extend((int(k0), None if int(b) == none else int(b), int(k1))
                                      for b, (k0, k1) in zip(row, bounds))
        return Chromagram(total)

    @classmethod
    def compute_batch(cls, clips, fs, frame_size=8192, num_harmonic=2, num_octave=2, num_bins=2, hop=None,
                      device=0):
        """Many clips in one launch -> list of Chromagram."""
        sums = get_engine(device).harmonic_energy_batch(clips, fs, frame_size, hop, num_harmonic, num_octave,
                                                        num_bins)
        return [Chromagram(s) for s in sums]
