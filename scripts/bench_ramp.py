"""Per-update completion times after a device sync: where do short runs (the driver's K=20) lose time?
Records an event on the caller stream after every update (= end of the value phase of update i) and the host time
at which the update's launches had been issued."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from porl_amd.agent.por import POR
from porl_amd.buffer.replay_buffer import PackedReplay
from porl_amd.util.synth import make_rows
from porl_amd import engine as E

S, A, H, L, B = 60, 2, 1024, 2, 1024
dev = torch.device("cuda", 0)
replay = PackedReplay(make_rows(1_000_000, S, A, seed=1000), S, A, dev, rank=0, world=1, seed=0)
args = SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=A, max_batch=B)
torch.manual_seed(0)
agent = POR(args, max_steps=1000, tau=0.9, alpha=10.0, device=dev)
agent.async_losses = True
agent.pipeline = os.environ.get("PIPE", "1") == "1"
K = int(os.environ.get("K", "300"))
losses = torch.zeros(K + 50, 8, device=dev)
spin = float(os.environ.get("SPIN", "50"))
if spin > 0:
    sa, sb = torch.randn(4096, 1024, device=dev), torch.randn(1024, 1024, device=dev)
    sc = torch.empty(4096, 1024, device=dev)
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < spin:
        for _ in range(20):
            E.gemm_f32(0, sa, sb, 4096, 1024, 1024, 1024, 1024, sc, 1024)
        torch.cuda.synchronize()
if os.environ.get("SPINMODE") == "mix":
    # GEMMs and 5-stream sweeps like the update's, queued without host syncs in between
    sa, sb = torch.randn(4096, 1024, device=dev), torch.randn(1024, 1024, device=dev)
    sc = torch.empty(4096, 1024, device=dev)
    bufs = [torch.zeros(4_200_000, device=dev) for _ in range(5)]
    for it in range(int(os.environ.get("SPINIT", "150"))):
        E.gemm_f32(0, sa, sb, 4096, 1024, 1024, 1024, 1024, sc, 1024)
        E.adam_ema(bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], 1e-4, it + 1, ema_beta=0.005)
        E.gemm_f32(0, sa, sb, 4096, 1024, 1024, 1024, 1024, sc, 1024)
    torch.cuda.synchronize()
if os.environ.get("SPINMODE") == "mem":
    bufs = [torch.zeros(21_000_000, device=dev) for _ in range(5)]
    for it in range(int(os.environ.get("SPINIT", "1000"))):
        E.adam_ema(bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], 1e-4, it + 1, ema_beta=0.005)
    torch.cuda.synchronize()
if os.environ.get("SPINMODE") in ("two", "twosmall"):
    # two streams at once, no agent: short-block GEMMs on both (and sweeps on the second), or only tiny kernels on both
    small = os.environ["SPINMODE"] == "twosmall"
    sa, sb = torch.randn(4096, 1024, device=dev), torch.randn(1024, 1024, device=dev)
    sc, sc2 = torch.empty(4096, 1024, device=dev), torch.empty(4096, 1024, device=dev)
    bufs = [torch.zeros(4_200_000 if not small else 4096, device=dev) for _ in range(5)]
    side = torch.cuda.Stream(device=dev)
    M = 64 if small else 4096
    for it in range(int(os.environ.get("SPINIT", "150"))):
        E.gemm_f32(0, sa, sb, M, 1024, 1024, 1024, 1024, sc, 1024, tile=3)
        with torch.cuda.stream(side):
            E.gemm_f32(0, sa, sb, M, 1024, 1024, 1024, 1024, sc2, 1024, tile=3)
            E.adam_ema(bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], 1e-4, it + 1, ema_beta=0.005)
    torch.cuda.synchronize()
if os.environ.get("SPINMODE") in ("sigpp", "sigload"):
    # signal ping-pong between two streams (the pipelined update's ordering mechanism) with tiny kernels, or with the
    # update-sized GEMMs / sweeps in between ("sigload")
    import ctypes as C
    from porl_amd import _native as N
    lib = N.lib()
    sA, sB = C.c_void_p(), C.c_void_p()
    N.check(lib.porl_signal_create(C.byref(sA)), "sig"); N.check(lib.porl_signal_create(C.byref(sB)), "sig")
    load = os.environ["SPINMODE"] == "sigload"
    sa, sb = torch.randn(4096, 1024, device=dev), torch.randn(1024, 1024, device=dev)
    sc, sc2 = torch.empty(4096, 1024, device=dev), torch.empty(4096, 1024, device=dev)
    bufs = [torch.zeros(2_200_000 if load else 4096, device=dev) for _ in range(5)]
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    for it in range(1, int(os.environ.get("SPINIT", "100")) + 1):
        if load:
            E.gemm_f32(0, sa, sb, 4096, 1024, 1024, 1024, 1024, sc, 1024, tile=3)
        if it > 1:
            N.check(lib.porl_signal_wait_ge(sB, it - 1, C.c_void_p(main.cuda_stream)), "wait")
        E.adam_ema(bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], 1e-4, it, ema_beta=0.005)
        N.check(lib.porl_signal_write(sA, it, C.c_void_p(main.cuda_stream)), "write")
        with torch.cuda.stream(side):
            N.check(lib.porl_signal_wait_ge(sA, it, C.c_void_p(side.cuda_stream)), "wait")
            if load:
                E.gemm_f32(0, sa, sb, 4096, 1024, 1024, 1024, 1024, sc2, 1024, tile=3)
            N.check(lib.porl_signal_write(sB, it, C.c_void_p(side.cuda_stream)), "write")
            if load:
                E.gemm_f32(0, sa, sb, 2048, 1024, 1024, 1024, 1024, sc2, 1024, tile=3)
    torch.cuda.synchronize()
if os.environ.get("SPINMODE") == "agent":
    torch.manual_seed(1)
    scratch = POR(args, max_steps=1000, tau=0.9, alpha=10.0, device=dev)
    scratch.async_losses = True
    scratch.pipeline = agent.pipeline
    sl = torch.zeros(8, device=dev)
    for it in range(int(os.environ.get("SPINIT", "100"))):
        scratch._engine.set_stats(sl); scratch.update_from_replay(replay, B)
    scratch.flush()
    torch.cuda.synchronize()
if os.environ.get("MAIN_PRIORITY"):     # run the update with a high-priority caller stream (the side stream stays normal)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(os.environ["MAIN_PRIORITY"])))
WARM = int(os.environ.get("WARM", "5"))
for i in range(WARM):
    agent._engine.set_stats(losses[i % 5]); agent.update_from_replay(replay, B)
if os.environ.get("NOSYNC", "0") != "1":
    torch.cuda.synchronize()
if float(os.environ.get("SLEEP", "0")) > 0:
    time.sleep(float(os.environ["SLEEP"]))
    for i in range(int(os.environ.get("WARM2", "0"))):
        agent._engine.set_stats(losses[i % 5]); agent.update_from_replay(replay, B)
    torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
host = []
t0 = time.perf_counter()
ev[0].record()
for i in range(K):
    agent._engine.set_stats(losses[5 + i]); agent.update_from_replay(replay, B)
    ev[i + 1].record()
    host.append(time.perf_counter() - t0)
agent.flush()
torch.cuda.synchronize()
tot = time.perf_counter() - t0
ts = [ev[0].elapsed_time(e) * 1e3 for e in ev[1:]]
print("total %.3f ms for %d updates = %.1f/s" % (tot * 1e3, K, K / tot))
prev = 0.0
for i in range(min(K, 30)):
    print("upd %3d  value-phase end at %8.1f us (+%6.1f)   host issued at %8.1f us" % (i, ts[i], ts[i] - prev, host[i] * 1e6))
    prev = ts[i]
for a, b in ((30, 100), (100, 200), (200, 300)):
    if b <= K:
        print("updates %d-%d: %.1f us each; host %.1f us each" % (a, b, (ts[b - 1] - ts[a - 1]) / (b - a), (host[b - 1] - host[a - 1]) * 1e6 / (b - a)))
print("last value-phase end %.1f us, flush+sync end %.1f us" % (ts[-1], tot * 1e6))
