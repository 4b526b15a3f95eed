#!/bin/bash
# The headline of bench.py in the three ways it can be started on one GPU: plain, plain with the collective forced (RCCL communicator of one rank),
# under torch.distributed.run with one rank.  Median ms per step of each (same box, back to back, twice).
pick() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-28s value %.4g ms/step %.5f kernel_ms %.5f collective %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('collective')))" "$1"; }
for i in 1 2; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --headline-only 2>/dev/null | pick plain
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --headline-only --force-collective 2>/dev/null | pick force-collective
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$i bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --headline-only 2>/dev/null | pick torch.distributed.run
done
