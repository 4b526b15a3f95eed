"""One process per GPU: rank discovery and process-group bring-up shared by bench.py, corpus.py and stream.py.

The reference is single-process: frames / clips / time shards are independent units, so the data path has no
collective and the job ends with ONE all_gather of 12-vectors.  With one rank that gather is a no-op and is skipped --
unless the caller asks for it (`--force-collective`, or a launch through `torch.distributed.run --nproc-per-node 1`), in
which case the communicator is created and the collective runs over the one rank: the same RCCL code path a multi-GPU
launch takes, exercised on a one-GPU box."""
import os
import sys
import socket


def rank_world_local():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def under_launcher():
    """True inside a torch.distributed.run / torchrun worker (also a one-rank one)."""
    return "TORCHELASTIC_RUN_ID" in os.environ


def wants_collective(world, force=False):
    return world > 1 or bool(force) or under_launcher()


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_group(backend, device=None):
    """init_process_group over the env:// rendezvous; a bare one-rank process (no launcher) gets a loopback rendezvous of
    its own.  `device` (a cuda torch.device) binds the communicator to that GPU at once (RCCL comm init happens here, not in
    the first collective).  Call it BEFORE the process touches the GPU in any other way."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise RuntimeError("WORLD_SIZE > 1 without MASTER_PORT: launch with torch.distributed.run")
        os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("LOCAL_RANK", "0")
    # RCCL prints its version banner (library path, HIP / ROCm versions, host name) to STDOUT when the communicator is
    # created; the drivers' contract is ONE JSON line there.  File descriptor 1 points at stderr while the group comes up.
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        if device is not None and device.type == "cuda":
            dist.init_process_group(backend, device_id=device)
            import torch
            t = torch.zeros(1, device=device)
            dist.all_reduce(t)          # (the first collective: whatever the library still has to say)
            torch.cuda.synchronize(device)
        else:
            dist.init_process_group(backend)
    finally:
        sys.stdout.flush()
        try:   # the banner sits in the C library's buffer when stdout is a pipe: out with it while descriptor 1 is still stderr
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.dup2(saved, 1)
        os.close(saved)
    return dist
