"""GEMM micro-benchmark through the C ABI: one NT problem of M x N x K per tile configuration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from porl_amd import engine as E

dev = "cuda"
M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 1024, 1024))]
modes = {"NT": 0, "NN": 1, "TN": 2}
torch.manual_seed(0)
names = {0: "128x128/8w", 1: "128x64/8w", 2: "64x128/4w", 3: "64x64/4w", 4: "128x64/4w", 5: "128x128/4w"}
for mname, mode in modes.items():
    A = torch.randn((M, K) if mode < 2 else (K, M), device=dev)
    B = torch.randn((N, K) if mode == 0 else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    for tile in (0, 5, 1, 4, 3):
        for _ in range(3):
            E.gemm_f32(mode, A, B, M, N, K, A.shape[1], B.shape[1], C, N, tile=tile)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        n = 20
        t0.record()
        for _ in range(n):
            E.gemm_f32(mode, A, B, M, N, K, A.shape[1], B.shape[1], C, N, tile=tile)
        t1.record(); torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 1e3 / n
        print(f"{mname} {M}x{N}x{K} tile {names[tile]:11s}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF")
