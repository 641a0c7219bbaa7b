"""GEMM micro-benchmark on the encoder's shapes (huge M, small K): every tile configuration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from porl_amd import engine as E

dev = "cuda"
R1, R2 = 256 * 5760, 512 * 1440
cases = [(R1, 192, 96), (R1, 96, 192), (R2, 192, 384), (R2, 384, 192)]
from porl_amd import _native as NN
def knob(key, t):
    NN.check(NN.lib().porl_tune_set_ptr(key, NN.ptr(t) if t is not None else None))
VARIANTS = os.environ.get("VARIANTS", "plain").split(",")
names = {0: "128x128", 1: "128x64", 2: "64x128", 3: "64x64"}
torch.manual_seed(0)
for M, N, K in cases:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev)
    C = torch.empty(M, N, device=dev)
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
    cst = torch.empty((M + 31) // 32 * 2 * N, device=dev)
    for variant in VARIANTS:
      knob(b"gemm_a_scale", sc if "apro" in variant else None); knob(b"gemm_a_shift", sh if "apro" in variant else None)
      knob(b"gemm_resid", C if "resid" in variant else None); knob(b"gemm_cstat", cst if "cstat" in variant else None)
      print("variant", variant)
      for tile in (1, 3) if "apro" in variant else (0, 1, 2, 3):
        E.prof_enable(True)
        for _ in range(5):
            E.gemm_f32(0, A, B, M, N, K, K, K, C, N, tile=tile)
        prof = E.prof_read()
        E.prof_enable(False)
        us = sum(p["total_ms"] for p in prof if p["name"].startswith("gemm")) * 1e3 / 5
        print(f"NT {M}x{N}x{K} tile {names[tile]:8s}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
    for k in (b"gemm_a_scale", b"gemm_a_shift", b"gemm_resid", b"gemm_cstat"):
        knob(k, None)
