"""How much does one dependent kernel launch cost on this stack?  Trivial kernels through the C ABI."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from porl_amd import engine as E
from porl_amd import _native as N
idx = torch.empty(256, dtype=torch.int64, device="cuda")
lib = N.lib()
st = N.current_stream_ptr()
def run(n):
    for i in range(n):
        lib.porl_sample_indices(100000, 256, 1, i, 0, N.ptr(idx), st)
for n in (200, 2000):
    run(50); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); run(n); e1.record(); th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{n} trivial dependent launches: GPU {e0.elapsed_time(e1)*1e3/n:.2f} us each, host issue {th*1e6/n:.2f} us each")
# same under a captured graph
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    st2 = N.current_stream_ptr()
    for i in range(3): lib.porl_sample_indices(100000, 256, 1, i, 0, N.ptr(idx), st2)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        st3 = N.current_stream_ptr()
        for i in range(200): lib.porl_sample_indices(100000, 256, 1, i, 0, N.ptr(idx), st3)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
g.replay(); torch.cuda.synchronize()
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"graph of 200 trivial launches: {e0.elapsed_time(e1)*1e3/200:.2f} us each")
