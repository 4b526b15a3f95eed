#!/bin/bash
# the crashing test under rocgdb: native backtrace of the faulting thread
export OPENBLAS_NUM_THREADS=1 OMP_NUM_THREADS=1
cat > /tmp/gdbcmds <<'G'
set pagination off
set confirm off
handle all nostop noprint pass
handle SIGSEGV stop print nopass
handle SIGABRT stop print nopass
run
bt 30
thread apply all bt 14
quit
G
timeout 600 rocgdb -q -batch -x /tmp/gdbcmds --args python3 -m pytest tests/test_corpus.py -m gpu -x -q 2>&1 | grep -v "^\[New Thread\|^\[Thread\|warning:\|^\[Switching" | head -400
