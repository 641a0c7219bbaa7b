"""Does a small HBM-bound kernel run beside a chip-filling fp32-MFMA GEMM launched on another stream?
Stream A: the 4 x 1024^3 grouped GEMM (256 blocks of 128x128, one per CU).  Stream B: an Adam sweep over the policy
group (33 MB), issued right after A.  Reports B's duration alone, beside A, and with a high-priority stream B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from porl_amd import engine as E

dev = torch.device("cuda", 0)
M = N = K = 1024
A = [torch.randn(M, K, device=dev) for _ in range(4)]
Bm = [torch.randn(N, K, device=dev) for _ in range(4)]
Cm = [torch.empty(M, N, device=dev) for _ in range(4)]
n = 1_173_624 // 4 * 4
p, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
v = v.abs()


TILE = [0, 4096]


def big():
    # one M x 1024 x 1024 problem: M = 4096 -> 256 tiles of 128x128, one per CU (tile 1: 512 blocks of 128x64)
    E.gemm_f32(0, Abig, Bm[0], TILE[1], N, K, K, K, Cbig, N, tile=TILE[0])


Abig = torch.randn(4096, K, device=dev)
Cbig = torch.empty(4096, N, device=dev)


def small():
    E.adam_ema(p, g, m, v, None, 1e-4, 3)


def timed(fn, stream, reps=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        for _ in range(5):
            fn()
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


sa = torch.cuda.Stream()
for tile, prio, rows, pad in ((0, 0, 4096, 0), (1, 0, 4096, 0), (1, 0, 2048, 0), (3, 0, 1024, 0), (3, 0, 4096, 0), (1, 0, 2048, 60000)):
    TILE[0], TILE[1] = tile, rows
    E.tune_set("gemm_lds_pad", pad)
    sb = torch.cuda.Stream(priority=prio)
    bm, bn = ((128, 128), (128, 64), (64, 128), (64, 64))[tile]
    print(f"GEMM tile {bm}x{bn}, M={rows}: {rows // bm * (1024 // bn)} blocks, extra LDS {pad} B")
    print(f"stream B priority {prio}:  GEMM alone {timed(big, sa):7.1f} us   Adam alone {timed(small, sb):7.1f} us")
    # interleaved issue: A then B, 50 times; total wall time vs the sum
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
    t0 = time.perf_counter()
    for i in range(50):
        with torch.cuda.stream(sa):
            big()
        with torch.cuda.stream(sb):
            ev[i][0].record(sb)
            small()
            ev[i][1].record(sb)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 50 * 1e6
    dur = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[25]
    print(f"   both streams: {wall:7.1f} us per pair (serial would be the sum); median Adam launch-to-finish {dur:7.1f} us")
