"""hipMalloc / first touch / hipFree time by size, straight through libamdhip64 (fresh process per call recommended)."""
import ctypes, sys, time
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hip.hipDeviceSynchronize()
sizes = [float(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 24, 32, 48]
p0 = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(p0), 1 << 20); hip.hipDeviceSynchronize()
for gib in sizes:
    n = int(gib * (1 << 30))
    out = []
    for rep in range(2):
        p = ctypes.c_void_p()
        t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), n); hip.hipDeviceSynchronize(); t1 = time.perf_counter()
        hip.hipMemset(p, 0, n); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
        hip.hipMemset(p, 0, n); hip.hipDeviceSynchronize(); t3 = time.perf_counter()
        hip.hipFree(p); hip.hipDeviceSynchronize(); t4 = time.perf_counter()
        out.append("rc %d alloc %.3f memset1 %.3f memset2 %.3f free %.3f" % (rc, t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    print("%5.1f GiB: %s | again: %s" % (gib, out[0], out[1]))
