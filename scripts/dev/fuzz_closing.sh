#!/bin/bash
# the differential-fuzz tools with fresh seeds on the closing tree (minutes of NumPy oracle per tool)
O=${1:-gpurun_out/fuzz_closing}
mkdir -p $O
timeout 1200 python3 tests/tools/fuzz_if0.py 70 611 2>&1 | grep -v amdgpu | tail -4 > $O/fuzz_if0.txt
timeout 900 python3 tests/tools/fuzz_he_prime.py 80 612 2>&1 | grep -v amdgpu | tail -4 > $O/fuzz_he_prime.txt
timeout 900 python3 tests/tools/fuzz_batch.py 8 613 2>&1 | grep -v amdgpu | tail -4 > $O/fuzz_batch.txt
FUZZ_DEFAULT_MODE=1 timeout 1200 python3 tests/tools/fuzz_esacf.py 40 614 2>&1 | grep -v amdgpu | tail -4 > $O/fuzz_esacf.txt
tail -2 $O/*.txt
