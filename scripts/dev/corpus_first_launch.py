"""When does the main context's first kernel of a 4096-clip ESACF call reach the queue (mpx_launch_count moves), with and
without a Prime-multiF0 call started at the same moment on another context?  Clip layout cold (fresh engines per trial)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import corpus
fs, n, dev = 22050, 4096, "cuda:0"
block = corpus.synth_block(n, fs, 2.0, 1024, 0, 1, synth_device=dev)
clips = block[0][1]
torch.cuda.synchronize()
for trial, with_prime in enumerate((False, True, True, True, True)):
    main, other = cd.Engine(0), cd.Engine(0)
    small = clips[:64].contiguous()
    main.esacf_batch(small, fs, int(fs * 46.4 / 1000.0))   # plans, kernels loaded
    other.prime_multif0_batch(small, fs)
    c0 = main.launch_count()
    t_first = [None]
    stop = threading.Event()
    t0 = time.perf_counter()
    def watch():
        while not stop.is_set():
            if main.launch_count() != c0:
                t_first[0] = time.perf_counter() - t0
                return
            time.sleep(0.00005)
    w = threading.Thread(target=watch); w.start()
    tp = [0.0]
    def prime():
        a = time.perf_counter(); other.prime_multif0_batch(clips, fs); tp[0] = time.perf_counter() - a
    th = threading.Thread(target=prime) if with_prime else None
    if th: th.start()
    a = time.perf_counter()
    cd.Engine._pack(clips)
    tpack = time.perf_counter() - a
    main.esacf_batch(clips, fs, int(fs * 46.4 / 1000.0))
    te = time.perf_counter() - a
    if th: th.join()
    stop.set(); w.join()
    print("prime beside: %-5s  Engine._pack alone took %.2f ms;  ESACF's first kernel enqueued after %.2f ms, its call returned after %.1f ms, prime's after %.1f ms" % (
        with_prime, 1e3 * tpack, 1e3 * (t_first[0] or -1), 1e3 * te, 1e3 * tp[0]), flush=True)
    main.close(); other.close()
