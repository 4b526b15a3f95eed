"""Method 4 -- drop-in for reference prime_multif0.py:18-91, HIP-backed."""
from .chromagram import Chromagram
from .engine import get_engine
from .multipitch import Multipitch


class MultipitchPrimeMultiF0(Multipitch):
    def __init__(
        self,
        audio_path,
        num_harmonic=1,
        num_octave=2,
        harmonic_multiples_elim=5,
        harmonic_elim_runs=2,
        fs=None,
        device=0,
        note_names="unicode",
    ):
        super().__init__(audio_path, fs=fs, device=device, note_names=note_names)
        self.num_harmonic = num_harmonic
        self.num_octave = num_octave
        self.harmonic_elim_runs = harmonic_elim_runs
        self.harmonic_multiples_elim = harmonic_multiples_elim

    @staticmethod
    def display_name():
        return "Prime-multiF0 (Camacho, Kaver-Oreamuno)"

    @staticmethod
    def method_number():
        return 4

    def compute_pitches(self, display_plot_frame=-1):
        total = get_engine(self.device).prime_multif0(
            self._samples(), self.fs, self.num_harmonic, self.num_octave, self.harmonic_multiples_elim,
            self.harmonic_elim_runs, note_names=self.note_names)
        return Chromagram(total)

    @classmethod
    def compute_batch(cls, clips, fs, num_harmonic=1, num_octave=2, harmonic_multiples_elim=5,
                      harmonic_elim_runs=2, device=0, note_names="unicode"):
        sums = get_engine(device).prime_multif0_batch(clips, fs, num_harmonic, num_octave,
                                                      harmonic_multiples_elim, harmonic_elim_runs,
                                                      note_names=note_names)
        return [Chromagram(s) for s in sums]
