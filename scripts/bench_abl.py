import sys, os, ctypes as C, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
M, N, K = 4096, 1024, 1024
names = {0: "real", 6: "no MFMA (memory pipeline only)"}
for tile in (0, 3):
    for mode, mname in ((0, "NT"), (2, "TN")):
        A = torch.randn((M, K) if mode < 2 else (K, M), device="cuda")
        B = torch.randn((N, K) if mode == 0 else (K, N), device="cuda")
        Cm = torch.empty(M, N, device="cuda")
        for abl in (0,):
            lib = C.CDLL(os.path.join(HERE, "_abl", f"libabl_t{tile}_{abl}.so"))
            f = lib.abl_gemm
            f.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            call = lambda: f(mode, M, N, K, A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], Cm.data_ptr(), N, st)
            for _ in range(3): call()
            torch.cuda.synchronize()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(20): call()
            t1.record(); torch.cuda.synchronize()
            us = t0.elapsed_time(t1) * 1e3 / 20
            print(f"tile{tile} {mname} abl={abl} {names[abl]:22s}: {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF")



import numpy as np
lib = C.CDLL(os.path.join(HERE, "_abl", "libabl_stamp_0.so"))
f = lib.abl_gemm
f.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); Cm = torch.empty(M, N, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(200):
    f(0, M, N, K, A.data_ptr(), K, B.data_ptr(), K, Cm.data_ptr(), N, st)
torch.cuda.synchronize()
buf = np.zeros(16 * 256, dtype=np.uint64)
lib.abl_stamps.argtypes = [C.c_void_p, C.c_int]
lib.abl_stamps(buf.ctypes.data, buf.size)
s = buf.reshape(-1, 16).astype(np.float64)
nk = K // 32
print(f"main loop: median {np.median(s[:,2]-s[:,1])*10:.0f} ns; cycles/K-tile {np.median(s[:,4])/nk:.0f}; clock {np.median(s[:,4]/(s[:,2]-s[:,1]))*0.1:.3f} GHz; setup {np.median(s[:,1]-s[:,0])*10:.0f} ns; epilogue {np.median(s[:,3]-s[:,2])*10:.0f} ns")
print("cycles per K-tile by k-group (ideal 1024 each):", [round(float(np.median(s[:, 6 + g]) / nk)) for g in range(4)])
