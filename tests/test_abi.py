"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/mpx.h declares.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from chord_detection_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mpx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from chord_detection_amd import _lib
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libmpx_hip.so does not export " + n
        assert n in _lib.SIGNATURES, "no ctypes signature for " + n
    assert set(_lib.SIGNATURES) == set(names)


def test_version_and_frame_math(lib):
    assert lib.mpx_abi_version() == 6
    assert lib.mpx_dev_knobs() == 0      # the release build: no result-changing environment switches compiled in
    assert lib.mpx_num_frames(44100, 1023, 1023) == 44
    assert lib.mpx_num_frames(44100, 8192, 8192) == 6
    assert lib.mpx_num_frames(8391680, 4096, 1024) == 8192   # BASELINE shape
    assert lib.mpx_num_frames(0, 4096, 1024) == 0
    assert lib.mpx_num_frames(10, 4096, 1024) == 1
    assert lib.mpx_num_frames(10, 0, 1) == -1
    assert lib.mpx_num_frames(10, 4, 8) == -1


def test_fails_loudly_without_gpu(lib):
    if lib.mpx_device_count() > 0:
        pytest.skip("a GPU is present")
    import chord_detection_amd as cd
    with pytest.raises(RuntimeError):
        cd.Engine(0)
    assert b"HIP device" in lib.mpx_last_error(None)


def test_host_api_surface():
    import chord_detection_amd as cd
    assert list(cd.METHODS.keys())[:2] == [1, 2]
    assert cd.MultipitchESACF.display_name() == "ESACF (Tolonen, Karjalainen)"
    assert cd.MultipitchHarmonicEnergy.display_name() == "Harmonic Energy (Stark, Plumbley)"
    assert cd.MultipitchESACF.method_number() == 1 and cd.MultipitchHarmonicEnergy.method_number() == 2
    with pytest.raises(ValueError):
        class Dup(cd.Multipitch):  # duplicate registration, multipitch.py:15-20
            @staticmethod
            def method_number():
                return 2
    import numpy as np
    with pytest.raises(ValueError):
        cd.MultipitchHarmonicEnergy(np.zeros((2, 8)), fs=22050)
    obj = cd.MultipitchESACF((np.zeros(100, dtype=np.float32), 22050))
    assert obj.ham_samples == 1023
    assert cd.MultipitchESACF((np.zeros(100, dtype=np.float32), 44100)).ham_samples == 2046
