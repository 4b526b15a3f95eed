"""Method registry and base class, mirroring reference multipitch.py:6-44.

Differences from the reference, all additive:
  * besides a path, the constructor accepts in-memory audio: a 1-D array with
    the keyword `fs=`, or an (x, fs) tuple -- loading/resampling is outside the
    accelerated path;
  * `device=` selects the GPU (default 0);
  * `note_names=` ("unicode" default | "ascii") says how the librosa the reference runs with spells sharps in
    `hz_to_note` -- with unicode names the reference's `chromagram[note] += v` loses C#, D#, F#, G#, A#
    (chromagram.py:19-29; include/mpx.h MPX_NOTES_*).  Methods 1, 3, 4; method 2 never calls hz_to_note.

Interface mirror of sevagh/chord-detection's chord_detection/multipitch.py (MIT License, Copyright (c) Sevag
Hanssian): the plug-in API -- the `METHODS` registry filled by `__init_subclass__`, the registration error and the four
abstract hooks (`display_name`, `method_number`, `compute_pitches`, `__init__`) -- is the reference's on purpose, so that
a method class written for the reference registers here unchanged; the constructor body is new.
"""
from abc import ABCMeta, abstractmethod
from collections import OrderedDict
from pathlib import Path

import numpy

METHODS = OrderedDict()


class Multipitch(object):
    __metaclass__ = ABCMeta

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        method_num = cls.method_number()
        if method_num in METHODS.keys():
            raise ValueError(
                "Method number {0} already registered as {1} in {2}".format(
                    method_num, METHODS[method_num], METHODS
                )
            )
        METHODS[cls.method_number()] = cls

    @abstractmethod
    def __init__(self, audio_path, fs=None, device=0, note_names="unicode"):
        if note_names not in ("unicode", "ascii"):
            raise ValueError("note_names must be 'unicode' or 'ascii'")
        self.note_names = note_names
        if isinstance(audio_path, tuple) and len(audio_path) == 2:
            x, self.fs = numpy.asarray(audio_path[0]), audio_path[1]
            self.clip_name = "<array>"
        elif isinstance(audio_path, numpy.ndarray):
            if fs is None:
                raise ValueError("fs= is required when passing samples instead of a path")
            x, self.fs = audio_path, fs
            self.clip_name = "<array>"
        else:
            from . import audio
            self.clip_name = Path(audio_path).name
            pcm = audio.load_pcm16(audio_path)
            if pcm is not None:
                # a mono PCM_16 file at the methods' rate (what gen_test_clips.py writes and librosa.load reads back as
                # int16 / 32768, multipitch.py:24-30): the int16 samples go to the device as they are and are scaled there
                # (include/mpx.h "PCM_16 input"); `self.x` is the float32 view the reference's attribute holds, made on demand
                from .engine import Pcm16
                self._pcm16, self.fs = Pcm16(pcm[0]), pcm[1]
                self._x = None
                self.device = device
                return
            x, self.fs = audio.load(audio_path)
        if len(x.shape) != 1:
            raise ValueError("Only 1D numpy ndarrays are supported")
        self.x = numpy.ascontiguousarray(x, dtype=numpy.float32)
        self.device = device

    _pcm16 = None

    @property
    def x(self):
        if self._x is None and self._pcm16 is not None:
            self._x = self._pcm16.float32()
        return self._x

    @x.setter
    def x(self, value):
        self._x, self._pcm16 = value, None   # samples assigned by the caller replace the file's

    def _samples(self):
        """what compute_pitches hands the engine: the file's int16 samples when there are any, else `self.x`"""
        return self._pcm16 if self._pcm16 is not None else self._x

    @abstractmethod
    def compute_pitches(self):
        pass

    @staticmethod
    @abstractmethod
    def display_name():
        raise ValueError("unimplemented")

    @staticmethod
    @abstractmethod
    def method_number():
        raise ValueError("unimplemented")
