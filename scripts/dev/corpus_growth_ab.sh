#!/bin/bash
# The corpus group whose clip layout is new to the contexts (and whose workspaces grow) inside the timed call -- the bench's
# sequence -- in fresh processes: Iterative-F0 / Prime-multiF0 threads started at once (rounds 3-5) against waiting for the main
# context's stream to have work queued (corpus._start_side, round 6).
for k in 1 2 3 4 5; do
  timeout 200 python3 scripts/dev/corpus_growth_trial.py nowait 2>&1 | grep "timed group"
  timeout 200 python3 scripts/dev/corpus_growth_trial.py wait 2>&1 | grep "timed group"
done
