"""chord_detection_amd -- MI355X-native drop-in for the hot path of
sevagh/chord-detection (reference chord_detection/__init__.py:1-7 exports)."""
from .esacf import MultipitchESACF
from .harmonic_energy import MultipitchHarmonicEnergy
from .iterative_f0 import MultipitchIterativeF0
from .prime_multif0 import MultipitchPrimeMultiF0
from .multipitch import METHODS, Multipitch
from .chromagram import Chromagram, detect_key
from .engine import Engine, Pcm16, get_engine, device_count, pinned_empty

__all__ = ["MultipitchESACF", "MultipitchHarmonicEnergy", "MultipitchIterativeF0", "MultipitchPrimeMultiF0", "METHODS", "Multipitch", "Chromagram", "detect_key",
           "Engine", "Pcm16", "get_engine", "device_count", "pinned_empty"]
