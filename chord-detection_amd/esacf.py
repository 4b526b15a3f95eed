"""Method 1 -- drop-in for reference esacf.py:16-91, HIP-backed."""
from .chromagram import Chromagram
from .engine import get_engine
from .multipitch import Multipitch


class MultipitchESACF(Multipitch):
    def __init__(
        self,
        audio_path,
        ham_ms=46.4,
        k=0.67,
        n_peaks_elim=6,
        peak_thresh=0.1,
        peak_min_dist=10,
        enhance_mode="librosa010",
        hop=None,
        fs=None,
        device=0,
        note_names="unicode",
    ):
        super().__init__(audio_path, fs=fs, device=device, note_names=note_names)
        self.ham_samples = int(self.fs * ham_ms / 1000.0)
        self.k = k  # kept for signature parity; the reference never forwards it (esacf.py:53,95-96)
        self.n_peaks_elim = n_peaks_elim
        self.peak_thresh = peak_thresh
        self.peak_min_dist = peak_min_dist
        self.enhance_mode = enhance_mode
        self.hop = hop

    @staticmethod
    def display_name():
        return "ESACF (Tolonen, Karjalainen)"

    @staticmethod
    def method_number():
        return 1

    def compute_pitches(self, display_plot_frame=-1):
        total = get_engine(self.device).esacf(
            self._samples(), self.fs, self.ham_samples, self.hop, self.n_peaks_elim, self.peak_thresh,
            self.peak_min_dist, self.enhance_mode, note_names=self.note_names)
        return Chromagram(total)

    @classmethod
    def compute_batch(cls, clips, fs, ham_ms=46.4, n_peaks_elim=6, peak_thresh=0.1, peak_min_dist=10,
                      enhance_mode="librosa010", hop=None, device=0, note_names="unicode"):
        frame = int(fs * ham_ms / 1000.0)
        sums = get_engine(device).esacf_batch(clips, fs, frame, hop, n_peaks_elim, peak_thresh, peak_min_dist,
                                              enhance_mode, note_names=note_names)
        return [Chromagram(s) for s in sums]
