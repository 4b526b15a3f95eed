"""CPU: the unit-root tables of the Iterative-F0 summary-spectrum kernels (csrc/mpx_if0_tables.hpp, used by if0_plan) against the
entry-by-entry long double formulas of rounds 1-5, bit for bit, under AddressSanitizer (g++; no GPU, no HIP).  The shortcut the
header takes -- the lower half of W_NF copied from W_2NF -- first shipped reading past the end of W_2NF."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "mpx_if0_tables.hpp"
int main() {
    int bad = 0;
    for (int NF = 8; NF <= 16384; NF *= 2) {
        std::vector<double> tw0(2 * (size_t)NF), twn0(2 * (size_t)NF + 2);
        for (int j = 0; j < NF; ++j) {
            const long double ang = -2.0L * M_PIl * j / (long double)NF;
            tw0[2 * j] = (double)cosl(ang);
            tw0[2 * j + 1] = (double)sinl(ang);
        }
        for (int k = 0; k <= NF; ++k) {
            const long double ang = -2.0L * M_PIl * k / (long double)(2 * NF);
            twn0[2 * k] = (double)cosl(ang);
            twn0[2 * k + 1] = (double)sinl(ang);
        }
        // exactly-sized heap blocks: an access past either end is an AddressSanitizer report
        std::vector<double> tw(2 * (size_t)NF, -7.0), twn(2 * (size_t)NF + 2, -7.0);
        mpx::if0_unit_roots(NF, tw.data(), twn.data());
        const bool ok = !std::memcmp(tw.data(), tw0.data(), tw0.size() * 8) && !std::memcmp(twn.data(), twn0.data(), twn0.size() * 8);
        std::printf("NF %5d %s\n", NF, ok ? "equal" : "DIFFERENT");
        bad += !ok;
    }
    return bad;
}
"""


def test_if0_unit_roots_are_the_entry_by_entry_tables(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.run([gxx, "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-I", os.path.join(ROOT, "chord-detection_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("equal") == 12 and "DIFFERENT" not in r.stdout, r.stdout
