"""Per-kernel register / scratch / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python3 scripts/kernel_resources.py            # compiles csrc/*.hip for gfx950 (device only), prints the table
    python3 scripts/kernel_resources.py --json f   # and writes it as JSON

tests/test_kernel_resources.py asserts on the same table (scratch-free kernels stay scratch-free)."""
import json
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chord-detection_amd", "csrc")
UNITS = ["mpx_he.hip", "mpx_esacf.hip", "mpx_prime.hip", "mpx_if0.hip", "mpx_api.hip"]
KEYS = {"SGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch",
        "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds"}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*$", "", o.replace("void ", "")) for o in out[:len(names)]]


def unit_resources(unit, extra=()):
    with tempfile.TemporaryDirectory() as td:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only",
               "-Rpass-analysis=kernel-resource-usage", "-w", *extra, "-c", os.path.join(CSRC, unit), "-o", os.path.join(td, "o")]
        err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    rows, cur = {}, None
    for line in err.split("\n"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+(.*?): (\d+) \[-Rpass", line)
        if m and cur and m.group(1) in KEYS:
            rows[cur][KEYS[m.group(1)]] = int(m.group(2))
    names = list(rows)
    return {d: rows[n] for n, d in zip(names, demangle(names))}


def all_resources(units=UNITS, extra=()):
    with ThreadPoolExecutor(4) as ex:
        parts = list(ex.map(lambda u: unit_resources(u, extra), units))
    out = {}
    for p in parts:
        out.update(p)
    return out


if __name__ == "__main__":
    table = all_resources()
    for k, v in sorted(table.items()):
        print("%-64s vgpr %3d agpr %3d scratch %4d lds %6d occ %d" % (k[:64], v.get("vgpr", -1), v.get("agpr", -1), v.get("scratch", -1), v.get("lds", -1), v.get("occupancy", -1)))
    if "--json" in sys.argv:
        json.dump(table, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1, sort_keys=True)
