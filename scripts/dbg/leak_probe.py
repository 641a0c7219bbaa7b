import gc, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from types import SimpleNamespace
from porl_amd import _native as N
from porl_amd.engine import IqlEngine
dev = torch.device("cuda", 0)

def free():
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0]

lib = N.lib()
torch.zeros(1, device=dev)
f0 = free()
# 1. raw signal create/destroy
for _ in range(50):
    p = C.c_void_p(); assert lib.porl_signal_create(C.byref(p)) == 0; assert lib.porl_signal_destroy(p) == 0
f1 = free(); print("50 x signal create/destroy:", (f0 - f1) >> 10, "KiB")
# 2. engine create / destroy, no use
for _ in range(50):
    e = IqlEngine(24, 24, 128, 2, max_batch=64, device=dev); del e
f2 = free(); print("50 x engine create/drop:", (f1 - f2) >> 10, "KiB")
# 3. engine + signals()
for _ in range(50):
    e = IqlEngine(24, 24, 128, 2, max_batch=64, device=dev); e.signals(); del e
f3 = free(); print("50 x engine + signals:", (f2 - f3) >> 10, "KiB")
# 4. engine + side stream
for _ in range(50):
    e = IqlEngine(24, 24, 128, 2, max_batch=64, device=dev); e.side_stream(); del e
f4 = free(); print("50 x engine + side stream:", (f3 - f4) >> 10, "KiB")
# 5. full agent cycle
from porl_amd.agent.por import POR
from porl_amd.buffer.replay_buffer import PackedReplay
from porl_amd.util.synth import make_rows
S, A, H, B = 24, 2, 128, 64
rows = make_rows(4096, S, A, seed=3)
def cyc(pipeline, n_upd):
    replay = PackedReplay(rows, S, A, dev, rank=0, world=1, seed=0)
    agent = POR(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=2, layer_norm=False, action_size=A, max_batch=B), max_steps=100, tau=0.9, alpha=10.0, device=dev)
    agent.async_losses, agent.pipeline = True, pipeline
    for _ in range(n_upd): agent.update_from_replay(replay, B)
    agent.flush()
for name, pl, n in (("agent no update", False, 0), ("agent one-stream x4", False, 4), ("agent pipelined x4", True, 4)):
    cyc(pl, n); fa = free()
    for _ in range(25): cyc(pl, n)
    fb = free(); print("25 x", name, ":", (fa - fb) >> 10, "KiB")
