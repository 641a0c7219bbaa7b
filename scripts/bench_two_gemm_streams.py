"""Upper bound for the pipelined update: aggregate fp32-MFMA rate of independent 4 x 1024^3-sized products issued back to
back on ONE stream vs on TWO streams at once, per tile shape (0 128x128, 1 128x64, 2 64x128, 3 64x64) and LDS pad."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from porl_amd import engine as E

dev = torch.device("cuda", 0)
M, N, K = 4096, 1024, 1024
a = [torch.randn(M, K, device=dev) for _ in range(2)]
b = [torch.randn(N, K, device=dev) for _ in range(2)]
c = [torch.empty(M, N, device=dev) for _ in range(2)]
side = torch.cuda.Stream(device=dev)
flop = 2.0 * M * N * K


def run(tile, two, n=200):
    for rep in range(2):                           # first pass warms up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            E.gemm_f32(0, a[0], b[0], M, N, K, K, K, c[0], N, tile=tile)
            if two:
                with torch.cuda.stream(side):
                    E.gemm_f32(0, a[1], b[1], M, N, K, K, K, c[1], N, tile=tile)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    k = n * (2 if two else 1)
    return el / k * 1e6, k * flop / el / 1e12


pads = [int(x) for x in os.environ.get("PADS", "0,18432").split(",")]
tiles = [int(x) for x in os.environ.get("TILES", "2,3").split(",")]
for pad in pads:
    E.tune_set("gemm_lds_pad", pad)
    for tile in tiles:
        u1, t1 = run(tile, False)
        u2, t2 = run(tile, True)
        print(f"pad {pad:6d} tile {tile}: one stream {u1:6.1f} us/launch {t1:6.1f} TF | two streams {u2:6.1f} us/launch {t2:6.1f} TF aggregate", flush=True)
