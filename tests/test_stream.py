"""Long-stream driver (BASELINE configs[4]): frame partition with a sample halo, the world-size-2 gloo path on
CPU with the checker standing in for the engine (which also proves the warm-up argument on the restated
reference itself), and -- on the GPU -- the sharded job against the unsharded one and the oracle."""
import os
import socket
import sys
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chord_detection_amd  # noqa: E402,F401
from chord_detection_amd import stream  # noqa: E402

FS = 22050
FRAME = 8192


def test_shard_windows_tile_the_stream():
    for n in (0, 1, FRAME, 20 * FRAME + 17, 158760000):
        nf = stream.num_frames(n, FRAME)
        for world in (1, 2, 3, 8):
            wins = [stream.shard_window(n, FRAME, world, r) for r in range(world)]
            assert wins[0][0] == 0 and wins[-1][1] == nf
            for (f0, f1, s0, s1, skip), nxt in zip(wins, wins[1:] + [None]):
                if nxt is not None:
                    assert f1 == nxt[0]
                if f1 > f0:
                    assert s1 == min(n, f1 * FRAME) and s0 == max(0, f0 * FRAME - stream.WARMUP)
                    assert s0 % FRAME == 0 and skip == (f0 * FRAME - s0) // FRAME and skip <= stream.WARMUP // FRAME
    with pytest.raises(ValueError):
        stream.shard_window(10 * 3000, 3000, 2, 0)      # the warm-up must be whole frames


def test_synthetic_stream_is_a_function_of_the_position():
    a = stream.synth_stream(0, 30000, FS)
    b = stream.synth_stream(9000, 26000, FS)
    assert a.dtype == torch.float32 and a.shape == (30000,)
    assert torch.equal(a[9000:26000], b)                 # a rank synthesises exactly its own window
    assert 0.02 < float(a.abs().max()) < 1.0
    assert stream.segment_notes(3) == stream.segment_notes(3) != stream.segment_notes(4)


def _oracle_frames(x, fs, frame_size, device, **kw):
    from oracle import iterative_f0 as o_if0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return o_if0.iterative_f0_frames(np.asarray(x, dtype=np.float32), fs, frame_size=frame_size)[0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


N_CPU = 21 * FRAME + 1234   # 22 frames: rank 1 owns 11..21 and starts 8 frames (65536 samples) early, at frame 3


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = stream.synth_stream(0, N_CPU, FS).numpy()
    f0, f1, block = stream.run_stream_shard(lambda a, b: x[a:b], N_CPU, FS, rank, world, FRAME, compute=_oracle_frames)
    frames = stream.gather_frames(block, stream.num_frames(N_CPU, FRAME), world, rank)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), f0=f0, f1=f1, block=block, frames=frames)
    dist.destroy_process_group()


def test_two_ranks_shard_the_stream_and_gather(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert (int(r0["f0"]), int(r0["f1"]), int(r1["f0"]), int(r1["f1"])) == (0, 11, 11, 22)
    np.testing.assert_array_equal(r0["frames"], r1["frames"])
    assert r0["frames"].shape == (22, 12)
    np.testing.assert_array_equal(r0["frames"][:11], r0["block"])
    np.testing.assert_array_equal(r0["frames"][11:], r1["block"])
    # the sharded job equals the sequential one: rank 1 started from zero state 65536 samples before its frames
    x = stream.synth_stream(0, N_CPU, FS).numpy()
    whole = _oracle_frames(x, FS, FRAME, None)
    assert whole.shape == (22, 12) and np.abs(whole).sum() > 0
    np.testing.assert_allclose(r0["frames"], whole, rtol=1e-9, atol=1e-12)
    assert repr(stream.chroma_of(r0["frames"])) == repr(stream.chroma_of(whole))


@pytest.mark.gpu
def test_sharded_stream_matches_the_whole_stream_on_gpu():
    import chord_detection_amd as cd
    n = 60 * FRAME + 4321
    x = stream.synth_stream(0, n, FS, "cuda:0").cpu().numpy()
    eng = cd.get_engine(0)
    total, whole = eng.iterative_f0(x, FS, return_frames=True)
    assert whole.shape == (61, 12)
    for world in (2, 3, 8):
        blocks = [stream.run_stream_shard(lambda a, b: x[a:b], n, FS, r, world, FRAME)[2] for r in range(world)]
        got = np.concatenate(blocks, axis=0)
        np.testing.assert_allclose(got, whole, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(stream.chroma_of(whole).as_array(), total, rtol=1e-9)
    want = _oracle_frames(x[:12 * FRAME], FS, FRAME, None)
    np.testing.assert_allclose(whole[:12], want, rtol=1e-9, atol=1e-12)
    # other frame sizes divide the warm-up as well
    f0, f1, blk = stream.run_stream_shard(lambda a, b: x[a:b], 90 * 2048, FS, 1, 2, 2048)
    _, ref = eng.iterative_f0(x[:90 * 2048], FS, return_frames=True, frame_size=2048)
    np.testing.assert_allclose(blk, ref[f0:f1], rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
def test_time_shards_in_flight_on_one_gpu():
    """run_stream_rank: a rank's frames as two time shards on two contexts at once == the unsharded run."""
    import chord_detection_amd as cd
    n = 50 * FRAME + 999
    x_dev = stream.synth_stream(0, n, FS, "cuda:0")
    x = x_dev.cpu().numpy()
    _, whole = cd.get_engine(0).iterative_f0(x, FS, return_frames=True)
    for world, sub in ((1, 2), (2, 2), (1, 3)):
        rows = []
        for r in range(world):
            f0, f1, blk = stream.run_stream_rank(lambda a, b: x_dev[a:b], n, FS, r, world, FRAME, 0, sub=sub)
            assert (f0, f1) == stream.shard_window(n, FRAME, world, r)[:2]
            rows.append(blk)
        np.testing.assert_allclose(np.concatenate(rows), whole, rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
def test_drop_in_class_on_long_audio_runs_in_time_slices():
    """MultipitchIterativeF0.compute_pitches on audio whose front-end output exceeds the context's workspace cap (forced
    here by a 64 MiB cap: 30 frames need 137 MB) runs in time slices inside the library and returns what the one-piece call
    returns, bit for bit; run_stream_rank's one-call route (its own cap, set for the call and restored) does too."""
    import chord_detection_amd as cd
    n = 30 * FRAME + 77
    x = stream.synth_stream(0, n, FS, "cuda:0").cpu().numpy()
    eng = cd.get_engine(0)
    want = cd.MultipitchIterativeF0((x, FS)).compute_pitches()
    _, want_rows = eng.iterative_f0(x, FS, return_frames=True)
    before = eng.get_option("if0_workspace_bytes")
    try:
        eng.set_option("if0_workspace_bytes", 64 << 20)
        got = cd.MultipitchIterativeF0((x, FS)).compute_pitches()
    finally:
        eng.set_option("if0_workspace_bytes", before)
    np.testing.assert_array_equal(got.as_array(), want.as_array())
    assert repr(got) == repr(want)
    f0, f1, rows = stream.run_stream_rank(lambda a, b: x[a:b], n, FS, 0, 1, FRAME, 0, workspace_bytes=256 << 20)
    assert (f0, f1) == (0, 31) and eng.get_option("if0_workspace_bytes") == before
    np.testing.assert_array_equal(rows, want_rows)


@pytest.mark.gpu
def test_iterative_f0_at_44100_across_chunk_and_piece_boundaries(monkeypatch):
    """BASELINE configs[4] runs at 44.1 kHz.  The oracle filters the whole signal sequentially; the engine cuts it into
    chunks with a zero-state run-in, and the stream driver cuts a rank's frames into pieces once more.  40 frames
    (7.4 s): every frame against the oracle, in particular the ones on both sides of a chunk boundary and of a piece
    boundary; the summary spectra too."""
    import chord_detection_amd as cd
    from oracle import iterative_f0 as o_if0
    fs = 44100
    n = 40 * FRAME + 1234
    x = stream.synth_stream(0, n, fs, "cuda:0").cpu().numpy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want, Ut = o_if0.iterative_f0_frames(x, fs)
    assert want.shape == (41, 12) and np.abs(want).sum() > 0
    eng = cd.get_engine(0)
    w, rho = eng.iterative_f0_warmup(fs)
    assert w == 40960 and 0.99892 < rho < 0.99894          # the default chain; stream.WARMUP, the CPU tests' halo, is larger
    total, got = eng.iterative_f0(x, fs, return_frames=True)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=0)
    ut = eng.iterative_f0_spectra(x, fs)
    np.testing.assert_allclose(ut, Ut, rtol=1e-9, atol=1e-9 * np.abs(Ut).max())
    # time shards (2, 3 ranks) and pieces of 4 frames inside a rank: piece boundaries at frames 4, 8, ...
    monkeypatch.setattr(stream, "PIECE_BYTES", 4 * FRAME * 70 * 8)
    for world in (1, 2, 3):
        rows = [stream.run_stream_rank(lambda a, b: x[a:b], n, fs, r, world, FRAME, 0, sub=2)[2] for r in range(world)]
        sharded = np.concatenate(rows)
        np.testing.assert_allclose(sharded, want, rtol=1e-5, atol=0)
        np.testing.assert_allclose(sharded, got, rtol=1e-9, atol=1e-12)
    assert repr(stream.chroma_of(got)) == repr(stream.chroma_of(want))


@pytest.mark.gpu
def test_run_in_follows_the_slowest_pole():
    """The library derives the run-in from the parameters instead of assuming 0.999: a channel set that reaches higher
    has slower resonators and gets a longer run-in; one that cannot be sharded at all is refused."""
    import chord_detection_amd as cd
    eng = cd.get_engine(0)
    w0, r0 = eng.iterative_f0_warmup(22050)
    w1, r1 = eng.iterative_f0_warmup(44100, channels=80, zeta1=0.45)        # top channel near 20 kHz
    assert w0 == 40960 and r1 > r0 and w1 > w0 and w1 % 8192 == 0
    tail = lambda w: np.exp(-w * (1 - r1)) * ((w * (1 - r1)) ** 3 + 3 * (w * (1 - r1)) ** 2 + 6 * w * (1 - r1) + 6) / 6   # Q(4, u)
    assert tail(w1) <= 1e-13 < tail(w1 - 8192)
    n = w1 + 5 * FRAME
    x = stream.synth_stream(0, n, 44100, "cuda:0").cpu().numpy()
    kw = dict(channels=80, zeta1=0.45)
    _, whole = eng.iterative_f0(x, 44100, return_frames=True, **kw)
    for world in (2, 3):
        rows = [stream.run_stream_shard(lambda a, b: x[a:b], n, 44100, r, world, FRAME, **kw)[2] for r in range(world)]
        np.testing.assert_allclose(np.concatenate(rows), whole, rtol=1e-9, atol=1e-12)
    with pytest.raises(ValueError):
        eng.iterative_f0_warmup(2000000, channels=100, zeta1=0.75)          # top channel at 865 kHz: pole radius 0.999994,
                                                                            # 6.4 M samples of run-in (> 4 M): refused
    w2, r2 = eng.iterative_f0_warmup(768000, channels=100, zeta1=0.65)      # 298 kHz: radius 0.99998 -> 2.2 M samples: accepted
    assert 2_000_000 < w2 < 2_400_000 and w2 % 8192 == 0 and 0.99998 < r2 < 0.99999
    with pytest.raises(ValueError):
        eng.iterative_f0(np.zeros(9000, dtype=np.float32), 2000000, channels=100, zeta1=0.75)


@pytest.mark.gpu
def test_one_hour_stream_shard_count_invariance():
    """BASELINE configs[4] at full size: Iterative-F0 over ONE 1 h stream @44.1 kHz (19 380 frames).  Size-independent
    property: the frames do not depend on how many time shards (GPUs) the stream was cut into -- 1, 2, 3 and 8 shards,
    each started the run-in (40960 samples for this chain) early from zero state, agree to 1e-9, and the run is reproducible."""
    import time
    fs, secs = 44100, 3600.0
    n = int(round(secs * fs))
    x = stream.synth_stream(0, n, fs, "cuda:0")
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    ref = None
    for world in (1, 2, 3, 8):
        t0 = time.perf_counter()
        rows = [stream.run_stream_rank(lambda a, b: x[a:b], n, fs, r, world, FRAME, 0, sub=2, note_names="ascii")[2]
                for r in range(world)]
        got = np.concatenate(rows)
        print("1 h stream, %d shard(s): %.2f s" % (world, time.perf_counter() - t0))
        assert got.shape == (19380, 12) and np.isfinite(got).all() and (got.sum(axis=1) > 0).all()   # ASCII names: every voice counts
        if ref is None:
            ref = got
            again = stream.run_stream_rank(lambda a, b: x[a:b], n, fs, 0, 1, FRAME, 0, sub=2, note_names="ascii")[2]
            np.testing.assert_array_equal(again, ref)
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12)
    assert repr(stream.chroma_of(got)) == repr(stream.chroma_of(ref))
