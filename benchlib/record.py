"""The record bench.py prints: the full one goes to a file, the driver's line is the compact one (<= 6 KB)."""
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPACT_LIMIT = 6000   # bytes of the last stdout line (tests/test_gpu_bench_contract.py asserts it)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _short(s, n=120):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "~"


def _r(v, sig=6):
    """numbers to `sig` significant digits (the full record keeps every bit)"""
    if isinstance(v, float):
        return float("%.*g" % (sig, v)) if math.isfinite(v) else None
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_from", "kernel", "kernel_ms", "compulsory_bytes",
             "wasted_traffic_ratio")


def _cpu_compact(cb):
    if not cb:
        return None
    r = _pick(cb, ("value", "unit", "cores", "kind"))
    r["sample"] = _short(cb.get("sample", ""), 100)
    if isinstance(cb.get("all_cores"), dict):
        r["all_cores"] = _pick(cb["all_cores"], ("value", "cores"))
    return r


def compact_record(out, full_path=None):
    """The driver's line: the contract's keys, the headline roofline + cpu_baseline, and one short entry per workload."""
    rec = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                      "vs_baseline", "dtype", "data", "engine", "ms_per_step_repeats", "value_one_in_flight",
                      "ms_per_step_one_in_flight", "value_steps_only", "ms_per_step_steps_only", "gather_ms"))
    cfg = out.get("config", {})
    rec["config"] = _pick(cfg, ("frames_per_gpu", "fft", "hop", "fs", "repeats", "repeat_statistic", "batches_in_flight",
                                "distinct_input_signals"))
    rec["config"]["workload"] = _short(cfg.get("workload", ""), 160)
    roof = out.get("roofline", {})
    rec["roofline"] = _pick(roof, ROOF_KEYS + ("bytes_per_frame", "frames_per_launch", "step_ms_hip_events"))
    if "secondary" in roof:
        rec["roofline"]["secondary"] = _pick(roof["secondary"], ("bound", "achieved", "peak", "unit", "frac"))
    if out.get("cpu_baseline"):
        rec["cpu_baseline"] = _cpu_compact(out["cpu_baseline"])
    wl = {}
    for name, w in (out.get("workloads") or {}).items():
        e = _pick(w, ("value", "unit", "scaling", "value_definition", "value_warm", "value_first_pass", "value_cold",
                      "value_one_call", "value_three_in_flight", "value_kernel_only", "value_with_streaming_synthesis", "value_without_synthesis",
                      "oracle_spot_check"))
        ms = w.get("ms_per_batch", 1e3 * w["wall_s"] if "wall_s" in w else None)
        if ms is not None:
            e["ms"] = ms
        r = w.get("roofline") or {}
        e.update(_pick(r, ("kernel", "kernel_ms", "bound", "frac", "traffic", "compulsory_bytes", "wasted_traffic_ratio")))
        if "hbm_frac_whole_path" in w:
            e["hbm_frac_whole_path"] = w["hbm_frac_whole_path"]
        if "traffic_vs_sample_bytes" in w:   # Iterative-F0: every kernel's HBM bytes / the samples' own bytes (the fp64 hand-off)
            e["traffic_vs_sample_bytes"] = w["traffic_vs_sample_bytes"]
        if (w.get("roofline") or {}).get("traffic_vs_compulsory_plus_intermediate") is not None:
            e["traffic_vs_compulsory_plus_intermediate"] = w["roofline"]["traffic_vs_compulsory_plus_intermediate"]
        km = w.get("kernels_ms") or w.get("kernels_ms_total")
        if km:
            e["kernels_ms"] = {k: v for k, v in sorted(km.items(), key=lambda kv: -kv[1])[:6]}
        cb = w.get("cpu_baseline")
        if cb:
            e["cpu"] = _pick(cb, ("value", "cores"))
            if isinstance(cb.get("all_cores"), dict):
                e["cpu"]["all_cores"] = _pick(cb["all_cores"], ("value", "cores"))
        e["workload"] = _short((w.get("config") or {}).get("workload", ""), 110)
        wl[name] = e
    if wl:
        rec["workloads"] = wl
    if full_path:
        rec["full_record"] = full_path
    return _r(rec)


def write_full_record(out, path):
    """the uncut record: `path`, and a copy under gpurun_out/ when that directory exists (it travels back from the GPU box)"""
    written = None
    for p in (path, os.path.join(ROOT, "gpurun_out", os.path.basename(path))):
        try:
            if p != path and not os.path.isdir(os.path.dirname(p)):
                continue
            with open(p, "w") as f:
                json.dump(out, f)
            written = written or os.path.relpath(p, os.getcwd())
        except OSError:
            pass
    return written

