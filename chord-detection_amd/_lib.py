"""ctypes binding of include/mpx.h (the C-ABI shared library built from csrc/).

There is NO CPU fallback: if libmpx_hip.so is missing or no MI355X is visible,
every compute entry point raises.
"""
import ctypes as C
import os
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPX_LIB_PATH") or os.path.join(_PKG, "libmpx_hip.so")  # env override: A/B builds

MPX_OK, MPX_EINVAL, MPX_ENOMEM, MPX_EHIP, MPX_EUNSUPPORTED = 0, -1, -2, -3, -4
MPX_FLAG_F32 = 0x1
MPX_FLAG_DETERMINISTIC = 0x2
MPX_ENHANCE_LIBROSA010, MPX_ENHANCE_NOOP = 0, 1
MPX_NOTES_UNICODE, MPX_NOTES_ASCII = 0, 1
MPX_OPT_IF0_WORKSPACE_BYTES, MPX_OPT_HE_KERNEL = 1, 2
MPX_HE_KERNEL_AUTO, MPX_HE_KERNEL_WORKGROUP, MPX_HE_KERNEL_WAVE_PAIRS = 0, 1, 2
OPTIONS = {"if0_workspace_bytes": MPX_OPT_IF0_WORKSPACE_BYTES, "he_kernel": MPX_OPT_HE_KERNEL}
NOTE_NAMES = {"unicode": MPX_NOTES_UNICODE, "ascii": MPX_NOTES_ASCII}
STAGES = {"wfir": 0, "x_lo": 1, "x_hi": 2, "sacf": 3, "esacf": 4}


class HeParams(C.Structure):
    _fields_ = [("num_harmonic", C.c_int), ("num_octave", C.c_int), ("num_bins", C.c_int)]


class PrimeParams(C.Structure):
    _fields_ = [("num_harmonic", C.c_int), ("num_octave", C.c_int), ("harmonic_multiples_elim", C.c_int),
                ("harmonic_elim_runs", C.c_int), ("note_names", C.c_int)]


class If0Params(C.Structure):
    _fields_ = [("frame_size", C.c_int), ("power", C.c_double), ("channels", C.c_int), ("zeta0", C.c_double),
                ("zeta1", C.c_double), ("max_voices", C.c_int), ("tau_min", C.c_double), ("tau_max", C.c_double),
                ("tau_prec", C.c_double), ("Q", C.c_int), ("M", C.c_int), ("epsilon1", C.c_double),
                ("epsilon2", C.c_double), ("gamma", C.c_double), ("note_names", C.c_int)]


class EsacfParams(C.Structure):
    _fields_ = [("n_peaks_elim", C.c_int), ("peak_thresh", C.c_double),
                ("peak_min_dist", C.c_int), ("enhance_mode", C.c_int), ("note_names", C.c_int)]


_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_vp = C.c_void_p
_sp = C.POINTER(C.c_int16)

# name -> (restype, argtypes); every symbol include/mpx.h declares
SIGNATURES = {
    "mpx_abi_version": (C.c_int, []),
    "mpx_launch_count": (C.c_uint, [_vp]),
    "mpx_dev_knobs": (C.c_int, []),
    "mpx_device_count": (C.c_int, []),
    "mpx_create": (_vp, [C.c_int, C.c_int]),
    "mpx_destroy": (None, [_vp]),
    "mpx_last_error": (C.c_char_p, [_vp]),
    "mpx_synchronize": (C.c_int, [_vp]),
    "mpx_stream": (_vp, [_vp]),
    "mpx_set_option": (C.c_int, [_vp, C.c_int, C.c_int64]),
    "mpx_get_option": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int64)]),
    "mpx_num_frames": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "mpx_harmonic_energy": (C.c_int, [_vp, _fp, C.c_int64, C.c_int, C.POINTER(HeParams), C.c_int, C.c_int, _dp, _dp]),
    "mpx_harmonic_energy_batch": (C.c_int, [_vp, _fp, _ip, C.c_int, C.c_int, C.POINTER(HeParams), C.c_int, C.c_int, _dp]),
    "mpx_harmonic_energy_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.POINTER(HeParams), C.c_int, C.c_int, _vp, _vp, _vp]),
    "mpx_harmonic_energy_argmax": (C.c_int, [_vp, _fp, C.c_int64, C.c_int, C.POINTER(HeParams), C.c_int, C.c_int,
                                             C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mpx_esacf": (C.c_int, [_vp, _fp, C.c_int64, C.c_int, C.POINTER(EsacfParams), C.c_int, C.c_int, _dp, _dp]),
    "mpx_esacf_batch": (C.c_int, [_vp, _fp, _ip, C.c_int, C.c_int, C.POINTER(EsacfParams), C.c_int, C.c_int, _dp]),
    "mpx_esacf_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.POINTER(EsacfParams), C.c_int, C.c_int, _vp, _vp, _vp]),
    "mpx_prime_multif0": (C.c_int, [_vp, _fp, C.c_int64, C.c_int, C.POINTER(PrimeParams), _dp]),
    "mpx_prime_multif0_batch": (C.c_int, [_vp, _fp, _ip, C.c_int, C.c_int, C.POINTER(PrimeParams), _dp]),
    "mpx_iterative_f0": (C.c_int, [_vp, _fp, C.c_int64, C.c_int, C.POINTER(If0Params), _dp, _dp]),
    "mpx_iterative_f0_batch": (C.c_int, [_vp, _fp, _ip, C.c_int, C.c_int, C.POINTER(If0Params), _dp]),
    "mpx_iterative_f0_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.POINTER(If0Params), _vp, _vp, _vp]),
    "mpx_prime_multif0_dev": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.POINTER(PrimeParams), _vp, _vp]),
    "mpx_iterative_f0_warmup": (C.c_int, [_vp, C.c_int, C.POINTER(If0Params), C.POINTER(C.c_int64), _dp]),
    "mpx_iterative_f0_spectra": (C.c_int, [_vp, _fp, C.c_int64, C.c_int, C.POINTER(If0Params), _dp]),
    "mpx_iterative_f0_periodicity": (C.c_int, [_vp, _dp, C.c_int64, C.c_int, C.c_int, C.POINTER(If0Params), _dp]),
    "mpx_iterative_f0_periodicity_voices": (C.c_int, [_vp, _dp, C.c_int64, C.c_int, C.c_int, C.POINTER(If0Params), _dp, _dp, _dp]),
    "mpx_harmonic_energy_pcm16": (C.c_int, [_vp, _sp, C.c_int64, C.c_int, C.POINTER(HeParams), C.c_int, C.c_int, _dp, _dp]),
    "mpx_esacf_pcm16": (C.c_int, [_vp, _sp, C.c_int64, C.c_int, C.POINTER(EsacfParams), C.c_int, C.c_int, _dp, _dp]),
    "mpx_prime_multif0_pcm16": (C.c_int, [_vp, _sp, C.c_int64, C.c_int, C.POINTER(PrimeParams), _dp]),
    "mpx_iterative_f0_pcm16": (C.c_int, [_vp, _sp, C.c_int64, C.c_int, C.POINTER(If0Params), _dp, _dp]),
    "mpx_esacf_stage": (C.c_int, [_vp, C.c_int, _fp, C.c_int64, C.c_int, C.POINTER(EsacfParams), C.c_int, C.c_int, _dp]),
    "mpx_set_remez_taps": (C.c_int, [_vp, C.c_int, _dp]),
    "mpx_test_gaussian_fit": (C.c_int, [_dp, _dp, C.c_int, _dp]),
    "mpx_test_pow067": (C.c_int, [_dp, C.c_int, _dp]),
    "mpx_test_lm_div_sqrt": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int, _dp, _dp]),
    "mpx_timer_begin": (C.c_int, [_vp, _vp]),
    "mpx_timer_end": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "mpx_host_alloc": (_vp, [C.c_size_t]),
    "mpx_host_free": (None, [_vp]),
    "mpx_profile_begin": (C.c_int, [_vp]),
    "mpx_profile_end": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "mpx_esacf_fit_stats": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
}

_lib = None
ABI_VERSION = 6   # include/mpx.h MPX_ABI_VERSION


def _share_torch_hip_runtime():
    """torch ships its own libamdhip64 with the same SONAME as the system one.  Two HIP runtimes in a process do not
    both see the GPU, so if torch is installed but not imported yet, its copy is loaded first (a few ms, no
    `import torch`): our library then binds to it, and a later `import torch` finds it already there."""
    import sys
    if "torch" in sys.modules or os.environ.get("MPX_NO_TORCH_HIP"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """Load libmpx_hip.so (once), sharing torch's HIP runtime when torch is installed (see above)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "chord_detection_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    # The library must be the ABI this binding was written for.  A library named by MPX_LIB_PATH (development A/B builds)
    # is held to the same check; only with MPX_ALLOW_OLD_ABI=1 is an earlier ABI (>= 2) accepted, and then the symbols it
    # lacks are simply not bound (calling one raises AttributeError naming it).
    abi = lib.mpx_abi_version() if hasattr(lib, "mpx_abi_version") else -1
    old_ok = bool(os.environ.get("MPX_LIB_PATH")) and os.environ.get("MPX_ALLOW_OLD_ABI") == "1" and 2 <= abi < ABI_VERSION
    if abi != ABI_VERSION and not old_ok:
        raise RuntimeError("%s: ABI version %d, this binding needs %d (rebuild: make -C chord-detection_amd/csrc; an older "
                           "A/B library loads only with MPX_LIB_PATH and MPX_ALLOW_OLD_ABI=1)" % (LIB_PATH, abi, ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        if old_ok and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
