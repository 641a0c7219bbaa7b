"""Run by tests/test_rccl_gpu.py in a process of its own: a process group of ONE rank on backend "nccl" (= RCCL on ROCm)
with the data-parallel exchange forced on (porl_amd.parallel.GradExchange(force=True)), so that the one GPU of the test
box executes the very calls the 8-GPU job of BASELINE config 4 makes: reduce_scatter_tensor / all_gather_into_tensor /
all_reduce, on the default communicator and on the policy group's own communicator, issued from the caller's stream and
from the engine's side stream, ordered against hipStreamWriteValue64 / hipStreamWaitValue64 operations and beside
chip-filling GEMM launches.  A SUM over one rank changes no number, so every mode must reproduce the plain single-GPU
update.  Prints one JSON line."""
import datetime
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    from types import SimpleNamespace
    from porl_amd.agent.por import POR
    from porl_amd.buffer.replay_buffer import PackedReplay
    from porl_amd.util.synth import make_rows, split_rows

    port = int(sys.argv[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev,
                            timeout=datetime.timedelta(seconds=90))
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "cases": []}

    def agent(S, H, B, force):
        torch.manual_seed(0)
        a = POR(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=2, layer_norm=False, action_size=2, max_batch=B),
                1000, 0.9, 10.0, device=dev)
        a._exchange.force = force
        return a

    # ---- 1. every exchange mode against the plain update, small network, explicit minibatches ----------------------
    S, H, B, K = 60, 128, 64, 4
    rows = torch.from_numpy(make_rows(K * B, S, 2, seed=9)).to(dev)

    def run(a):
        for k in range(K):
            s, r, sp, d, _ = split_rows(rows[k * B:(k + 1) * B], S, 2)
            a.por_residual_update(s, sp, r, d)
        sd = {k: v.clone() for k, v in a.state_dict().items()}
        a.v_optimizer.consolidate_state()
        m0 = a.v_optimizer.state_dict()["state"][0]["exp_avg"].clone()
        return sd, m0

    plain = agent(S, H, B, False)
    assert not plain._exchange.active
    want, want_m0 = run(plain)
    for exchange in ("reduce_scatter", "all_reduce"):
        for async_mode in (False, True):
            for pol_group in ((True, False) if async_mode else (False,)):
                a = agent(S, H, B, True)
                assert a._exchange.active and a._exchange.world_size == 1
                a.grad_exchange, a.async_losses, a.dp_policy_group = exchange, async_mode, pol_group
                got, m0 = run(a)
                err = max(float((got[k] - want[k]).abs().max()) for k in want)
                merr = float((m0 - want_m0).abs().max())
                two = getattr(a, "_exchange_pol", None) is not None
                out["cases"].append(dict(exchange=exchange, pipelined=async_mode, policy_group=pol_group,
                                         second_communicator=two, max_abs_param_err=err, max_abs_moment_err=merr))
                assert two == (async_mode and pol_group)
    # ---- 2. headline shape, pipelined, both communicators, rows drawn on the device: RCCL kernels beside the GEMMs ---
    S, H, B, K = 60, 1024, 1024, 40
    replay_rows = make_rows(50_000, S, 2, seed=3)
    res = {}
    for force in (False, True):
        a = agent(S, H, B, force)
        a.async_losses = True
        rp = PackedReplay(replay_rows, S, 2, dev, seed=5)
        hist = torch.zeros(K, 8, device=dev)
        for k in range(5):
            a.update_from_replay(rp, B)
        a.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            a._engine.set_stats(hist[k])
            a.update_from_replay(rp, B)
        a.flush(); torch.cuda.synchronize()
        res[force] = (time.perf_counter() - t0, hist[:, :3].cpu().numpy(), {k: v.clone() for k, v in a.state_dict().items()})
    assert np.isfinite(res[True][1]).all()
    out["headline"] = dict(updates=K, plain_ms_per_update=1e3 * res[False][0] / K, rccl_ms_per_update=1e3 * res[True][0] / K,
                           max_abs_param_err=max(float((res[True][2][k] - res[False][2][k]).abs().max()) for k in res[True][2]),
                           frac_params_beyond_2e6=float(sum(int(((res[True][2][k] - res[False][2][k]).abs() > 2e-6).sum())
                                                            for k in res[True][2]) / sum(v.numel() for v in res[True][2].values())),
                           max_rel_loss_err=float(np.abs(res[True][1][:, :2] / res[False][1][:, :2] - 1).max()))
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_WORLD1 " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
