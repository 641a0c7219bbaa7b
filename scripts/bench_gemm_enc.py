"""GEMM micro-benchmark on the encoder's shapes (huge M, small K): every tile configuration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from porl_amd import engine as E

dev = "cuda"
R1, R2 = 256 * 5760, 512 * 1440
cases = [(R1, 192, 96), (R1, 96, 192), (R2, 192, 384), (R2, 384, 192)]
names = {0: "128x128", 1: "128x64", 2: "64x128", 3: "64x64"}
torch.manual_seed(0)
for M, N, K in cases:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev)
    C = torch.empty(M, N, device=dev)
    for tile in (0, 1, 2, 3):
        E.prof_enable(True)
        for _ in range(5):
            E.gemm_f32(0, A, B, M, N, K, K, K, C, N, tile=tile)
        prof = E.prof_read()
        E.prof_enable(False)
        us = sum(p["total_ms"] for p in prof if p["name"].startswith("gemm")) * 1e3 / 5
        print(f"NT {M}x{N}x{K} tile {names[tile]:8s}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
