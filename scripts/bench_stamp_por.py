"""In-kernel clock stamps of the 128x128 GEMM on the POR hidden-layer shape (diagnostic build, scripts/gemm_abl.hip
with -DPORL_STAMP -DABL_TILE=0): cycles per K-tile by k-group, block life, epilogue."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "_abl", "libabl_stamp_t0.so"))
f = lib.abl_gemm
f.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
lib.abl_stamps.argtypes = [C.c_void_p, C.c_int]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 4096, 1024, 1024
A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); Cm = torch.empty(M, N, device="cuda")
for _ in range(50):
    f(0, M, N, K, A.data_ptr(), K, B.data_ptr(), K, Cm.data_ptr(), N, st)
torch.cuda.synchronize()
buf = np.zeros(16 * 4096, dtype=np.uint64)
lib.abl_stamps(buf.ctypes.data, buf.size)
s = buf.reshape(-1, 16).astype(np.float64)
s = s[s[:, 0] > 0]
nk = K // 32
groups = [float(np.median(s[:, 6 + g]) / nk) for g in range(4)]
print(f"NT {M}x{N}x{K}, tile 128x128, {len(s)} blocks stamped")
print(f"cycles per K-tile by k-group (ideal 1024 each): {[round(g) for g in groups]}  sum {sum(groups):.0f} (ideal 4096)")
print(f"block life (entry -> exit): {np.median(s[:,3]-s[:,0])*10:.0f} ns; after the main loop (C store): {np.median(s[:,3]-s[:,2])*10:.0f} ns")
