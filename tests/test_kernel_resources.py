"""CPU: register / scratch budget of the hot kernels, read from hipcc's own resource remarks for gfx950
(scripts/kernel_resources.py compiles csrc/*.hip device-only with -Rpass-analysis=kernel-resource-usage, ~15 s).

A kernel that DESIGN.md calls scratch-free must stay scratch-free: a spill in a frame loop is reloaded behind the
prefetch of the next frame (round 2 shipped he_wave_kernel with 24 B/lane that came from its ragged-frame path, and
if0_spectrum_dif_kernel with 184 B/lane, unnoticed)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.fixture(scope="module")
def table():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    import kernel_resources
    return kernel_resources.all_resources()


SCRATCH_FREE = [
    "mpx::he_wave_kernel<8, 4, false, true, 2146439166u, 1, false>",    # headline: every frame whole and aligned, 22 of 32 rows (HW_K2_44K)
    "mpx::he_wave_kernel<8, 4, false, true, 4294967295u, 1, false>",    # the same loader, window shapes / sample rates that need every row
    "mpx::he_wave_kernel<8, 4, false, false, 2146439166u, 1, false>",   # ragged / unaligned frames at 44.1 kHz
    "mpx::he_wave_kernel<7, 4, false, true, 4294967295u, 2, false>",    # 8192-sample frames (the reference's default), whole and aligned: two passes per frame
    "mpx::he_wave_kernel<6, 4, false, true, 4294967295u, 2, true>",     # ... a pair of waves per frame: the default for calls of >= 32 MiB (round 6: 20 B of scratch until the frame index became a scalar)
    "mpx::he_kernel<4096, 256, double>",
    "mpx::sacf_pfa_kernel<1>", "mpx::sacf_pfa_kernel<2>",
    "mpx::sacf_kernel<4096, false>", "mpx::sacf_kernel<4096, true>", "mpx::sacf_kernel<2048, true>",
    "mpx::bandsplit_kernel<false>", "mpx::bandsplit_kernel<true>",
    "mpx::pv_enhance_kernel<true, 4>", "mpx::sacf_split_kernel<8192, 512>", "mpx::sacf_rz_kernel<4096>", "mpx::scatter_kernel",
    "mpx::sacf_huge_kernel<512>", "mpx::pv_enhance_big_kernel<8>", "mpx::enhance_pick_big_kernel<512>",   # frames of odd length above 4096 / above 8192 samples
    "mpx::coopfit_kernel", "mpx::coopfit8_kernel",   # the cooperative fit kernels (four / eight fits to a wave)
    "mpx::peakfit_kernel<true>",                      # samples in LDS, fvec recomputed: every batch (the round-2 arrangement, 40 B of scratch, is a development-build option)
    "mpx::prime_wave_kernel<2048, 4>",                # Prime-multiF0, 2048-point chirp-z: a wave per SIMD, 512 registers each
    "mpx::prime_wave_kernel<1024, 7>",                # ... 1024 points, two items per wave at two waves per SIMD: 36 B of scratch until the lane-class masks of its ballots became scalars (round 6)
    "mpx::prime_pers_kernel<4096>",
    "mpx::if0_spectrum_split_kernel<8192, true, 1>",   # Iterative-F0 summary spectra at the default frame size, power 1
    # ... and every other instantiation a caller can reach through frame_size / power (iterative_f0.py:22-33); round 3 shipped
    # these with 96-324 bytes per lane of scratch under a four-waves-per-SIMD limit (if0_split_waves)
    "mpx::if0_spectrum_split_kernel<8192, false, 0>",
    "mpx::if0_spectrum_split_kernel<4096, true, 1>", "mpx::if0_spectrum_split_kernel<4096, false, 0>",
    "mpx::if0_spectrum_split_kernel<2048, true, 1>", "mpx::if0_spectrum_split_kernel<2048, false, 0>",
    "mpx::if0_spectrum_split_kernel<1024, true, 1>", "mpx::if0_spectrum_split_kernel<1024, false, 0>",
    "mpx::if0_frontend_kernel<false, 64>", "mpx::if0_frontend2_kernel<false>",
    "mpx::if0_frontend_kernel<true, 64>", "mpx::if0_frontend2_kernel<true>",     # time slices (MPX_OPT_IF0_WORKSPACE_BYTES)
    "mpx::if0_spectrum_blue_kernel<4096, 256>", "mpx::if0_spectrum_blue_kernel<8192, 512>", "mpx::if0_spectrum_blue2_kernel<512>",   # chirp-z frame sizes
    "mpx::if0_spectrum_blue4_kernel<512>", "mpx::if0_periodicity_kernel<true>",   # frame sizes 8193 ... 16384 (round 6)
]
# kernels that are known to spill, with the ceiling they must not grow past (bytes per lane).  Every ceiling is the value
# hipcc reports TODAY (ROCm 7.2), no headroom: a spill that grows by one slot fails here and has to be looked at (round 5
# shipped these with up to 4 bytes of slack and a 36-byte allowance for two kernels that no longer spill at all).
SCRATCH_CEILING = {
    "mpx::pv_enhance_kernel<true, 2>": 24,             # three workgroups per CU since round 5 (168 registers): 0.85 -> 0.66 ms per 8192 frames with the spill
    "mpx::if0_periodicity_kernel<false>": 108,                # held at four workgroups per CU (128 registers); fourteen loads in flight per lane in the range maxima: 1.59 -> 1.49 ms per 600 s WITH the spill
    "mpx::he_wave_kernel<8, 4, false, false, 4294967295u, 1, false>": 24,   # ragged / unaligned frames, every row: the loader with per-sample guards
    "mpx::he_wave_kernel<7, 4, false, false, 4294967295u, 2, false>": 32,
    "mpx::he_wave_kernel<6, 4, false, false, 4294967295u, 2, true>": 12,   # 8192-sample frames, ragged / unaligned (clips: the last frame of each)
}
# occupancy (waves per SIMD) the launch geometry of the host code counts on
OCCUPANCY = {
    "mpx::he_wave_kernel<8, 4, false, true, 2146439166u, 1, false>": 2,
    "mpx::he_wave_kernel<7, 4, false, true, 4294967295u, 2, false>": 2,
    "mpx::sacf_pfa_kernel<2>": 4,
    "mpx::if0_spectrum_split_kernel<8192, true, 1>": 4,
    "mpx::prime_wave_kernel<1024, 7>": 2,
    "mpx::prime_wave_kernel<2048, 4>": 1,
    "mpx::peakfit_kernel<true>": 2,
    "mpx::pv_enhance_kernel<true, 2>": 3,
    "mpx::if0_periodicity_kernel<false>": 4,                  # round 5 shipped it at 129 registers = 3 for a while: 17 % slower
}


def test_hot_kernels_stay_scratch_free(table):
    missing = [k for k in SCRATCH_FREE + list(SCRATCH_CEILING) + list(OCCUPANCY) if k not in table]
    assert not missing, "kernels not found in the compile remarks (renamed?): %s" % missing
    spilled = {k: table[k]["scratch"] for k in SCRATCH_FREE if table[k]["scratch"] != 0}
    assert not spilled, "scratch in kernels documented as scratch-free: %s" % spilled


def test_known_spills_do_not_grow(table):
    for k, cap in SCRATCH_CEILING.items():
        assert table[k]["scratch"] <= cap, (k, table[k])


def test_release_build_has_no_development_only_kernels(table):
    """if0_spectrum_kernel / if0_spectrum_dif_kernel and the prefetch variants of the split kernel are reachable only through
    dev_env knobs, which are constants in the release build: they are compiled under MPX_DEV_KNOBS only."""
    dead = [k for k in table if "if0_spectrum_kernel<" in k or "if0_spectrum_dif_kernel<" in k
            or ("if0_spectrum_split_kernel<" in k and not (k.endswith("true, 1>") or k.endswith("false, 0>")))
            or "prime_pers_kernel<1024>" in k or "prime_pers_kernel<2048>" in k        # round 3's Prime-multiF0 kernel where the wave kernel runs
            or ("prime_wave_kernel<" in k and k not in ("mpx::prime_wave_kernel<1024, 7>", "mpx::prime_wave_kernel<2048, 4>"))]
    assert not dead, dead


def test_occupancy_assumptions(table):
    for k, occ in OCCUPANCY.items():
        assert table[k]["occupancy"] >= occ, (k, table[k])
