"""Per-block timeline of the 64x64 GEMM on an encoder shape (diagnostic build with in-kernel clock stamps)."""
import sys, os, ctypes as C, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
os.makedirs(os.path.join(HERE, "_abl"), exist_ok=True)
so = os.path.join(HERE, "_abl", "libabl_stamp_t3.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DPORL_STAMP",
                    "-DABL_TILE=3", "-o", so, os.path.join(HERE, "gemm_abl.hip")], check=True)
lib = C.CDLL(so)
f = lib.abl_gemm
f.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
lib.abl_stamps.argtypes = [C.c_void_p, C.c_int]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for M, N, K in ((256 * 5760, 192, 96), (256 * 5760, 96, 192), (512 * 1440, 192, 384)):
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); Cm = torch.empty(M, N, device="cuda")
    for _ in range(3):
        f(0, M, N, K, A.data_ptr(), K, B.data_ptr(), K, Cm.data_ptr(), N, st)
    torch.cuda.synchronize()
    buf = np.zeros(16 * 4096, dtype=np.uint64)
    lib.abl_stamps(buf.ctypes.data, buf.size)
    s = buf.reshape(-1, 16).astype(np.float64)
    s = s[s[:, 0] > 0]
    nk = K // 32
    t0 = s[:, 0].min()
    print(f"{M}x{N}x{K}: blocks stamped {len(s)}; setup(entry->loop) {np.median(s[:,1]-s[:,0])*10:.0f} ns; "
          f"loop {np.median(s[:,2]-s[:,1])*10:.0f} ns ({np.median(s[:,4])/nk:.0f} cycles/K-tile, ideal 1024); "
          f"epilogue {np.median(s[:,3]-s[:,2])*10:.0f} ns; block life {np.median(s[:,3]-s[:,0])*10:.0f} ns; "
          f"first-wave entries span {(s[:,0].max()-t0)*10:.0f} ns", flush=True)
    print("   cycles per K-tile by k-group:", [round(float(np.median(s[:, 6 + g]) / nk)) for g in range(4)])
