"""Algorithmic work per unit of each kernel (DESIGN.md section 5 states the same figures) and the rooflines built on it."""
import json
import math
import os

from . import config as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_models(n, mh, channels=70, nf=8192):
    """name -> (bytes per unit, flops per unit, unit): compulsory HBM bytes and textbook flop counts."""
    lg = math.log2(max(n, 2))
    return {
        # ESACF, unit = frame of n samples, mh = (n-1)//2 lags
        "bandsplit_kernel": (4 * n + 16 * n, 90 * n, "frame"),            # fp32 in, (x_lo, x_hi) fp64 out; 12 all-pass + 13-tap FIR + 3 biquads
        "sacf_kernel": (16 * n + 8 * mh, 2 * 5 * n * lg + 40 * n, "frame"),   # two n-point complex DFTs + |.|^0.67 (log+exp) per bin
        "sacf_big_kernel": (16 * n + 8 * mh, 2 * 5 * n * lg + 40 * n, "frame"),
        "sacf_pfa_kernel": (16 * n + 8 * mh, 2 * 5 * n * lg + 40 * n, "frame"),   # the same algorithmic count whatever the engine
        "pv_enhance_kernel": (16 * mh, 2 * 6 * 2.5 * 2048 * 11, "frame"),  # two real vocoder rates x (4 STFT + 2 ISTFT) 2048-point real FFTs
        "peakpick_kernel": (8 * mh, 4 * mh, "frame"),
        "scatter_kernel": (96, 0, "frame"),
        # Iterative-F0, unit = sample (front end) or frame (spectra, search)
        # COMPULSORY bytes: the samples in once (4 B) and 12 doubles out per frame; what the three kernels hand
        # each other through HBM -- 8 B x channels per sample from the front end to the spectra, the 2 nf-bin summary
        # spectrum to the period search -- is INTERMEDIATE traffic and listed separately (INTERMEDIATE_BYTES below)
        "if0_frontend_kernel": (4, 110 * channels, "sample"),   # 17 IIR stages + 13-tap FIR per channel and sample
        "if0_spectrum_kernel": (0, channels * (2.5 * 2 * nf * math.log2(2 * nf) + nf), "frame"),
        "if0_periodicity_kernel": (96, 0, "frame"),
    }


# profile marks of the library (mpx_profile_*) -> kernels they cover, as the profiler names them
MARK_KERNELS = {"prime_kernel": ("prime_wave_kernel", "prime_pers_kernel", "prime_kernel"),   # wave per item (chirp-z of 1024 / 2048 points) | persistent workgroups | workgroup per item
                "if0_frontend_kernel": ("if0_frontend_kernel", "if0_frontend2_kernel"),   # pipelined | sequential (mpx_if0.hip)
                "if0_spectrum_kernel": ("if0_spectrum_split_kernel", "if0_spectrum_dif_kernel", "if0_spectrum_kernel"),
                "he_kernel": ("he_wave_kernel", "he_kernel", "he_blue_kernel"),
                "coopfit_kernel": ("coopfit_kernel", "coopfit8_kernel")}   # four or eight fits to a wave, chosen on the device (mpx_esacf.hip)
_TRAFFIC = None


def measured_traffic(pmc_workload, mark):
    """HBM bytes of the launches one profile mark covers, inside one workload of scripts/pmc_workloads.py, from the last
    PMC collection (profiles/traffic_latest.json; bench.py cannot run rocprofv3 on itself).  (bytes, note) or (None, None)."""
    global _TRAFFIC
    if _TRAFFIC is None:
        try:
            with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as fh:
                _TRAFFIC = json.load(fh)
        except Exception:
            _TRAFFIC = {}
    ks = _TRAFFIC.get("kernels", {}).get(pmc_workload, {})
    total, hit = 0.0, []
    for name, shapes in ks.items():
        base = name.split("<")[0]
        if any(base == p for p in MARK_KERNELS.get(mark, (mark,))):
            # a mark covers every launch shape of its kernels in the call (Prime-multiF0: one launch per chirp-z class)
            total += sum(sh["bytes_per_launch"] for sh in shapes)
            hit.append(name)
    if not hit:
        return None, None
    return total, "round %s PMC, %s: %s" % (_TRAFFIC.get("round"), pmc_workload, ", ".join(sorted(hit)))


def traffic_from(pmc_workload=None):
    """Short provenance of every `traffic` figure in the record: which committed PMC collection it was read from."""
    measured_traffic("he", "he_kernel")   # loads the file
    if not _TRAFFIC:
        return None
    return "committed PMC pass of round %s (profiles/traffic_latest.json <- %s), not this run" % (_TRAFFIC.get("round"), _TRAFFIC.get("source"))


def with_traffic(r, pmc_workload, mark, launches=1):
    """Fill roofline.traffic (bytes per call of the marked kernels x launches) and the ratio to the compulsory bytes."""
    if r is None:
        return r
    marks = mark.split("+")
    tot, notes = 0.0, []
    for m in marks:
        b, note = measured_traffic(pmc_workload, m)
        if b is None:
            return r
        tot += b
        notes.append(note)
    r["traffic"] = tot * launches
    r["traffic_from"] = traffic_from()
    r["traffic_note"] = "HBM-side bytes (calibrated factor x FETCH_SIZE + WRITE_SIZE, scripts/pmc_to_traffic.py) of these launches; " + "; ".join(notes)
    comp = r.get("bytes_per_unit", 0) * r.get("units_per_launch", 0)
    if r.get("compulsory_bytes"):
        comp = r["compulsory_bytes"]
    if comp:
        r["compulsory_bytes"] = comp
        r["wasted_traffic_ratio"] = r["traffic"] / comp
    ib = r.get("intermediate_bytes_per_unit", 0) * r.get("units_per_launch", 0)
    if ib:   # a kernel that hands data to the next one of its method: the measured bytes against compulsory + hand-off
        r["traffic_vs_compulsory_plus_intermediate"] = r["traffic"] / (comp + ib)
    return r


def he_kernel_name(f32):
    """The dominant kernel of the headline step: fp64 4096-sample frames run the wave-per-frame kernel (csrc/mpx_he_wave.hpp),
    fp32 the workgroup-per-frame one."""
    return "he_kernel<4096,256,float>" if f32 else "he_wave_kernel<8,4>"


def intermediate_bytes(name, channels=70, nf=8192):
    """Bytes per unit a kernel moves through HBM that are NOT compulsory: hand-offs between the kernels of one method."""
    return {"if0_frontend_kernel": 8 * channels,                      # writes [channel][t] fp64 for the spectra
            "if0_spectrum_kernel": 8 * nf * channels + 16 * nf,        # reads it back, writes the 2 nf-bin summary spectrum
            "if0_periodicity_kernel": 3 * 16 * nf}.get(name, 0)        # summary spectrum in, residual / detected spectra


def roofline_of(name, ms, units, model):
    """Both roofs for one kernel; `bound` is the one it sits closer to."""
    b, f, unit = model
    hbm = b * units / (ms * 1e-3) if ms > 0 else 0.0
    fl = f * units / (ms * 1e-3) if ms > 0 else 0.0
    hbm_frac, valu_frac = hbm / K.HBM_PEAK, fl / K.F64_PEAK
    if f and valu_frac >= hbm_frac:
        r = {"bound": "valu_f64", "achieved": fl / 1e12, "peak": K.F64_PEAK / 1e12, "unit": "TFLOP/s", "frac": valu_frac}
    else:
        r = {"bound": "hbm", "achieved": hbm / 1e9, "peak": K.HBM_PEAK / 1e9, "unit": "GB/s", "frac": hbm_frac}
    r.update({"kernel": name, "kernel_ms": ms, "units_per_launch": units, "unit_of_work": unit,
              "bytes_per_unit": b, "flops_per_unit": f, "hbm_frac": hbm_frac, "valu_f64_frac": valu_frac,
              "traffic": None})
    ib = intermediate_bytes(name)
    if ib:
        r["intermediate_bytes_per_unit"] = ib
        r["hbm_frac_with_intermediate"] = (b + ib) * units / (ms * 1e-3) / K.HBM_PEAK if ms > 0 else 0.0
    return r


FLOPS_PER_FIT_EVAL = 21 * 35   # one MINPACK function evaluation of a gaussian peak fit: 21 residuals x (exp ~30 flops + 5)


def fit_roofline(kms, stats):
    """The two gaussian-fit kernels together (peakfit_kernel runs the fits, coopfit_kernel finishes the runaway ones): work
    counted in MINPACK function evaluations (mpx_esacf_fit_stats) x 735 flops for the model evaluation alone -- the QR of
    the 21 x 3 jacobian and the 3 x 3 trust-region algebra of an iteration come on top and are not counted -- against the
    fp64 vector peak.  Bytes: the 21-sample window (168 B) in, 12 B out per fit: nothing."""
    ms = kms.get("peakfit_kernel", 0.0) + kms.get("coopfit_kernel", 0.0)
    fl = stats["evaluations"] * FLOPS_PER_FIT_EVAL / (ms * 1e-3) if ms > 0 else 0.0
    hbm = stats["fits"] * 180 / (ms * 1e-3) if ms > 0 else 0.0
    return {"bound": "valu_f64", "achieved": fl / 1e12, "peak": K.F64_PEAK / 1e12, "unit": "TFLOP/s", "frac": fl / K.F64_PEAK,
            "kernel": "peakfit_kernel+coopfit_kernel", "kernel_ms": ms, "units_per_launch": stats["evaluations"],
            "unit_of_work": "function evaluation", "bytes_per_unit": 0, "flops_per_unit": FLOPS_PER_FIT_EVAL,
            "compulsory_bytes": stats["fits"] * 180,   # a fit's 21-sample window in (168 B), centre + flag out (12 B)
            "hbm_frac": hbm / K.HBM_PEAK, "valu_f64_frac": fl / K.F64_PEAK, "traffic": None, "fits": stats["fits"],
            "evaluations_per_fit": stats["evaluations"] / max(stats["fits"], 1), "fits_finished_cooperatively": stats["parked"]}


def dominant(prof):
    name = max(prof, key=lambda k: prof[k][1])
    return name, prof[name][1] / max(prof[name][0], 1)
