"""GEMM micro-benchmark through the C ABI: one problem of M x N x K per tile configuration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from porl_amd import engine as E

dev = "cuda"
cases = [("NT", 4096, 1024, 60, 1), ("NT", 3072, 1024, 60, 1), ("NT", 4096, 1024, 1024, 1), ("TN", 1024, 60, 1024, 8), ("TN", 1024, 60, 1024, 16),
         ("NT", 1024, 60, 1024, 16), ("NT", 1024, 60, 1024, 8), ("TN", 60, 1024, 1024, 16), ("NN", 1024, 1024, 60, 1)]
modes = {"NT": 0, "NN": 1, "TN": 2}
names = {0: "128x128", 1: "128x64", 2: "64x128", 3: "64x64"}
torch.manual_seed(0)
for mname, M, N, K, sk in cases:
    mode = modes[mname]
    A = torch.randn((M, K) if mode < 2 else (K, M), device=dev)
    B = torch.randn((N, K) if mode == 0 else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    slab = torch.empty(sk * M * N, device=dev) if sk > 1 else None
    for tile in (0, 1, 2, 3):
        E.prof_enable(True)
        for _ in range(20):
            E.gemm_f32(mode, A, B, M, N, K, A.shape[1], B.shape[1], C, N, tile=tile, splitk=sk, slab=slab)
        prof = E.prof_read()
        E.prof_enable(False)
        us = sum(p["total_ms"] for p in prof if p["name"].startswith("gemm")) * 1e3 / 20
        print(f"{mname} {M}x{N}x{K} sk={sk:2d} tile {names[tile]:8s}: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF")
