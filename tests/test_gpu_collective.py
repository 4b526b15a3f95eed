"""RCCL on the one GPU of the test box: the multi-GPU code path of bench.py, scripts/run_corpus.py and scripts/run_stream.py
-- init_process_group("nccl", device_id=...) as the FIRST GPU call of the process, the job's all_gather / all_reduce,
destroy_process_group -- with ONE rank (one process per GPU, a single all_gather of 12-vectors at the end).
No scaling curve comes out of this; what it shows is that the communicator comes up and the collectives execute on an
MI355X, both under `--force-collective` and under a one-rank torch.distributed.run launch (how the driver starts N > 1)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return env


def _json_line(out):
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_headline_with_the_rccl_gather_on_one_rank(tmp_path):
    full = str(tmp_path / "full.json")
    cmd = [sys.executable, "bench.py", "--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--headline-only",
           "--full-json", full]
    forced = _json_line(subprocess.run(cmd + ["--force-collective"], cwd=ROOT, env=_env(), capture_output=True, text=True,
                                       timeout=900))
    with open(full) as fh:
        d = json.load(fh)
    assert d["engine"] == "hip" and d["n_gpus"] == 1
    assert d["config"]["collective"] == "nccl all_gather of [steps, 12] inside every timed repeat, 1 rank(s)"
    assert forced["value"] > 5e7
    launched = _json_line(subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
         "--master-port", str(_port())] + cmd[1:] + ["--gpus", "1"], cwd=ROOT, env=_env(), capture_output=True, text=True,
        timeout=900))
    with open(full) as fh:
        d = json.load(fh)
    assert d["config"]["collective"].startswith("nccl all_gather") and launched["n_gpus"] == 1
    # the gather of [200, 12] doubles once per 200-step repeat is noise next to the steps
    assert abs(launched["value"] - forced["value"]) <= 0.1 * forced["value"]


def test_corpus_and_stream_drivers_with_the_rccl_gather_on_one_rank():
    res = {}
    for flag in ([], ["--force-collective"]):
        out = subprocess.run([sys.executable, "scripts/run_corpus.py", "--clips", "48", "--chunk", "32", "--methods", "1,2,4"] + flag,
                             cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
        res[bool(flag)] = _json_line(out)
    assert res[False]["collective"] is None and res[True]["collective"] == "nccl all_gather over 1 rank(s)"
    for m in ("1", "2", "4"):   # the gathered block is the rank's block
        np.testing.assert_array_equal(res[False]["methods"][m]["mean_chroma"], res[True]["methods"][m]["mean_chroma"])
    res = {}
    for flag in ([], ["--force-collective"]):
        out = subprocess.run([sys.executable, "scripts/run_stream.py", "--seconds", "20", "--fs", "44100"] + flag,
                             cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
        res[bool(flag)] = _json_line(out)
    assert res[False]["collective"] is None and res[True]["collective"] == "nccl all_gather over 1 rank(s)"
    assert res[False]["chroma"] == res[True]["chroma"] and res[False]["key"] == res[True]["key"]
