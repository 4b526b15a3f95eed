"""CPU: engine.pinned_empty() must not free its pinned block while ANY array still looks at it -- plain-ndarray views
included (np.asarray / slices drop subclass attributes; the owner therefore hangs on the buffer NumPy keeps as base)."""
import ctypes
import gc

import numpy as np


class FakeLib:
    """mpx_host_alloc / mpx_host_free over malloc, recording the frees."""

    def __init__(self):
        self.libc = ctypes.CDLL(None)
        self.libc.malloc.restype = ctypes.c_void_p
        self.libc.malloc.argtypes = [ctypes.c_size_t]
        self.libc.free.argtypes = [ctypes.c_void_p]
        self.freed = []

    def mpx_host_alloc(self, nbytes):
        return self.libc.malloc(max(int(nbytes), 1))

    def mpx_host_free(self, ptr):
        self.freed.append(ptr)
        self.libc.free(ptr)


def test_views_keep_the_pinned_block_alive():
    from chord_detection_amd import engine
    lib = FakeLib()
    arr = engine.pinned_empty(1000, np.float32, lib=lib)
    assert arr.shape == (1000,) and arr.dtype == np.float32
    arr[:] = np.arange(1000, dtype=np.float32)
    plain = np.asarray(arr)            # what the engine's own _sig() does with its input
    sl = arr[10:20]
    other = arr.view(np.ndarray).reshape(10, 100)
    del arr
    gc.collect()
    assert lib.freed == []             # three views still alive
    assert plain[999] == 999.0 and sl[0] == 10.0 and other[9, 99] == 999.0
    del plain, sl
    gc.collect()
    assert lib.freed == []
    del other
    gc.collect()
    assert len(lib.freed) == 1         # freed exactly once, after the last view


def test_zero_length_and_dtype():
    from chord_detection_amd import engine
    lib = FakeLib()
    a = engine.pinned_empty(0, np.float64, lib=lib)
    assert a.size == 0
    del a
    gc.collect()
    assert len(lib.freed) == 1
