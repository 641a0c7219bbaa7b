"""Data-parallel update on the device: 2 ranks (both on cuda:0, gloo exchange of CUDA tensors — the box has
one GPU; RCCL needs one device per rank) against the single-process update on the concatenated minibatch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu
S, H, L, BL, STEPS = 60, 128, 2, 64, 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rows():
    from porl_amd.util.synth import make_rows
    return make_rows(2 * STEPS * BL, S, 2, seed=21)


def _agent(B):
    from types import SimpleNamespace
    from porl_amd.agent.por import POR
    torch.manual_seed(0)
    return POR(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=2, max_batch=B),
               1000, 0.9, 10.0, device=torch.device("cuda"))


def _worker(rank, world, port, out_dir, async_mode=False, exchange="reduce_scatter"):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import torch.distributed as dist
    from porl_amd.util.synth import split_rows
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    agent = _agent(BL)
    agent.grad_exchange = exchange
    agent.async_losses = async_mode                          # async: pipelined, policy phase + its exchange on the side stream
    rows = torch.from_numpy(_rows()).cuda()
    losses = []
    for k in range(STEPS):
        glob = rows[k * world * BL:(k + 1) * world * BL]
        local = glob[rank * BL:(rank + 1) * BL]              # this rank's shard of the global minibatch
        s, r, sp, d, _ = split_rows(local, S, 2)
        out = agent.por_residual_update(s, sp, r, d)
        losses.append(tuple(out[:2].tolist()) if async_mode else out)
    if async_mode:
        assert agent._engine._policy_done is not None        # the last policy phase is still on the side stream ...
    sd = agent.state_dict()                                  # ... and state_dict() completes it
    osd = agent.v_optimizer.state_dict()                     # (collective in reduce_scatter mode: moments are gathered)
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), losses=np.array(losses),
                 adam_m0=osd["state"][0]["exp_avg"].cpu().numpy(), adam_v2=osd["state"][2]["exp_avg_sq"].cpu().numpy(),
                 **{k: v.cpu().numpy() for k, v in sd.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["reduce_scatter", "all_reduce"])
@pytest.mark.parametrize("async_mode", [False, True])
def test_two_rank_update_equals_global_batch_update(tmp_path, async_mode, exchange):
    from porl_amd.util.synth import split_rows
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), async_mode, exchange), nprocs=world, join=True)
    got = np.load(tmp_path / "dp.npz")
    agent = _agent(world * BL)
    rows = torch.from_numpy(_rows()).cuda()
    for k in range(STEPS):
        s, r, sp, d, _ = split_rows(rows[k * world * BL:(k + 1) * world * BL], S, 2)
        loss = agent.por_residual_update(s, sp, r, d)
        if not async_mode:                                   # async mode reports per-rank loss shares
            np.testing.assert_allclose(got["losses"][k], loss, rtol=2e-6)
    for k, v in agent.state_dict().items():
        np.testing.assert_allclose(got[k], v.cpu().numpy(), atol=2e-6, err_msg=k)
    osd = agent.v_optimizer.state_dict()
    np.testing.assert_allclose(got["adam_m0"], osd["state"][0]["exp_avg"].cpu().numpy(), atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(got["adam_v2"], osd["state"][2]["exp_avg_sq"].cpu().numpy(), atol=1e-10, rtol=1e-4)


def test_bench_starts_its_own_ranks_from_the_bare_command():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (the driver's command form): the parent makes
    no GPU call, starts N rank processes itself and relays rank 0's JSON line.  Rehearsed with gloo (two ranks on the
    box's one GPU; RCCL needs a device per rank)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PORL_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--rows-per-gpu", "20000", "--no-roofline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"] == "gloo"
    assert out["config"]["global_batch"] == 2048 and out["scaling"] == "weak"
    assert np.isfinite([out["value"], out["final_losses"]["v_loss"], out["final_losses"]["g_loss"]]).all()


def test_bench_under_torch_distributed_run():
    """The driver's N > 1 command: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` — ranks come from the launcher's environment.  Rehearsed with gloo, two ranks
    on the box's one GPU."""
    import json
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PORL_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
                        "--gpus", "2", "--steps", "4", "--warmup", "2", "--rows-per-gpu", "20000", "--no-roofline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["config"]["global_batch"] == 2048
    assert np.isfinite([out["value"], out["final_losses"]["v_loss"], out["final_losses"]["g_loss"]]).all()


# ---- update -> vf_update in data-parallel mode (round-2 advisor finding: stale moments outside the rank's slice) --------
def _sorl(B):
    from types import SimpleNamespace
    from porl_amd.agent.sorl import SORL
    torch.manual_seed(0)
    return SORL(SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=2, max_batch=B),
                1000, 0.9, 3.0, device=torch.device("cuda"))


def _sorl_sequence(agent, rows, world, rank, per_rank):
    """update, vf_update, update, vf_update on successive global minibatches; `per_rank` picks this rank's share."""
    from porl_amd.util.synth import split_rows
    out = []
    for k in range(4):
        glob = rows[k * world * BL:(k + 1) * world * BL]
        local = glob[rank * BL:(rank + 1) * BL] if per_rank else glob
        s, r, sp, d, a = split_rows(local, S, 2)
        out.append(agent.update(s, a, r, sp, d)[0] if k % 2 == 0 else agent.vf_update(s, a, r, sp, d))
    return out


def _vf_worker(rank, world, port, out_dir, exchange, switch):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    agent = _sorl(BL)
    agent.grad_exchange = exchange
    rows = torch.from_numpy(_rows_n(4)).cuda()
    losses = _sorl_sequence(agent, rows, world, rank, True)
    if switch:                                               # change of exchange mode mid-run: moments must be gathered first
        agent.grad_exchange = "all_reduce"
        losses += _sorl_sequence(agent, rows, world, rank, True)
    agent.v_optimizer.consolidate_state()                    # every rank; afterwards state_dict() is local
    agent.policy_optimizer.consolidate_state()
    sd = {k: v.cpu().numpy() for k, v in agent.state_dict().items()}
    np.savez(os.path.join(out_dir, f"vf{rank}.npz"), losses=np.array(losses), **sd)
    if rank == 0:                                            # rank 0 ALONE: must not be a collective any more
        osd = agent.v_optimizer.state_dict()
        np.savez(os.path.join(out_dir, "vfopt.npz"), m0=osd["state"][0]["exp_avg"].cpu().numpy(),
                 v3=osd["state"][3]["exp_avg_sq"].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def _rows_n(n):
    from porl_amd.util.synth import make_rows
    return make_rows(2 * n * BL, S, 2, seed=33)


@pytest.mark.parametrize("exchange,switch", [("reduce_scatter", False), ("all_reduce", False), ("reduce_scatter", True)])
def test_update_then_vf_update_two_ranks_equals_single_process(tmp_path, exchange, switch):
    """SORL.update followed by SORL.vf_update (sorl.py:130-152) with sharded optimizer state: every rank must apply Adam
    with CURRENT moments, the ranks must stay replicas, and the result must equal the one-process run on the
    concatenated minibatches."""
    world = 2
    mp.spawn(_vf_worker, args=(world, _free_port(), str(tmp_path), exchange, switch), nprocs=world, join=True)
    g0, g1 = np.load(tmp_path / "vf0.npz"), np.load(tmp_path / "vf1.npz")
    agent = _sorl(world * BL)
    rows = torch.from_numpy(_rows_n(4)).cuda()
    want = _sorl_sequence(agent, rows, world, 0, False)
    if switch:
        want += _sorl_sequence(agent, rows, world, 0, False)
    np.testing.assert_allclose(g0["losses"], want, rtol=3e-6)
    for k, v in agent.state_dict().items():
        assert np.array_equal(g0[k], g1[k]), f"ranks diverged: {k}"
        np.testing.assert_allclose(g0[k], v.cpu().numpy(), atol=2e-6, err_msg=k)
    osd = agent.v_optimizer.state_dict()
    opt = np.load(tmp_path / "vfopt.npz")
    np.testing.assert_allclose(opt["m0"], osd["state"][0]["exp_avg"].cpu().numpy(), atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(opt["v3"], osd["state"][3]["exp_avg_sq"].cpu().numpy(), atol=1e-10, rtol=1e-4)
