#!/usr/bin/env python3
"""For `rocprofv3 --pmc FETCH_SIZE` (or --kernel-trace --stats): the three kernels of the 8192-sample Harmonic-Energy shape, each
over two 268 MB inputs in turn -- arg 1 = he_kernel option (0 one wave per frame, 2 a pair of waves per frame, 1 workgroup kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import chord_detection_amd as cd
opt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda", 0)
frames, N = 8196, 8192
g = torch.Generator(device="cpu"); g.manual_seed(1)
xs = [(0.3 * torch.randn(frames * N, generator=g)).to(dev) for _ in range(2)]
rows = torch.zeros((frames, 12), dtype=torch.float64, device=dev)
e = cd.Engine(0); e.set_option("he_kernel", opt)
for r in range(8):
    e.harmonic_energy_dev(xs[r & 1].data_ptr(), xs[0].numel(), 22050, N, N, rows.data_ptr(), None)
    e.synchronize()
print("done", opt)
