"""Host logic of the drop-in classes (no GPU): construction order / seeded init / state_dict keys equal
the reference's (golden files), optimizer + scheduler facades, replay sharding, and the data-parallel
exchange over gloo with world_size 2 (the oracle stands in for the device step)."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import load_golden, sub, REPO


def _args(S, H, L, ln=False, A=2):
    return SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=ln, feature_dim=256, action_size=A)


@pytest.mark.parametrize("name", ["por_s60_h64_b32", "por_s17_h48_l3_b50"])
def test_por_construction_matches_reference(name):
    from porl_amd.agent.por import POR
    z, meta = load_golden(name)
    torch.manual_seed(int(meta["seed_model"]))
    agent = POR(_args(int(meta["S"]), int(meta["H"]), int(meta["L"])), 1000, 0.9, 10.0)
    sd = agent.state_dict()
    assert list(sd.keys()) == [str(k) for k in z["keys"]]
    init = sub(z, "init/")
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), init[k]), k
    assert not any(p.requires_grad for p in agent.v_target.parameters())
    # parameters are views into three flat groups
    eng = agent._engine
    assert all(p.untyped_storage().data_ptr() == eng.params_vf.untyped_storage().data_ptr() for p in agent.vf.parameters())
    assert all(p.untyped_storage().data_ptr() == eng.params_pol.untyped_storage().data_ptr()
               for p in agent.goal_policy.parameters())


def test_sorl_construction_matches_reference():
    from porl_amd.agent.sorl import SORL
    z, meta = load_golden("sorl_s60_h64_b32")
    torch.manual_seed(int(meta["seed_model"]))
    agent = SORL(_args(60, 64, 2), 1000, 0.9, 3.0)
    sd = agent.state_dict()
    assert list(sd.keys()) == [str(k) for k in z["keys"]]
    init = sub(z, "init/")
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), init[k]), k


def test_load_state_dict_writes_through_to_flat_groups():
    from porl_amd.agent.por import POR
    torch.manual_seed(0)
    a = POR(_args(60, 64, 2), 1000, 0.9, 10.0)
    torch.manual_seed(1)
    b = POR(_args(60, 64, 2), 1000, 0.9, 10.0)
    b.load_state_dict(a.state_dict())
    assert torch.equal(a._engine.params_vf, b._engine.params_vf)
    assert torch.equal(a._engine.params_tgt, b._engine.params_tgt)
    assert torch.equal(a._engine.params_pol, b._engine.params_pol)


def test_cosine_schedule_equals_torch():
    from porl_amd.agent.por import POR
    a = POR(_args(8, 16, 1), 50, 0.9, 10.0, policy_lr=3e-4)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=3e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 50)
    for _ in range(130):                                   # past T_max: the reference's schedule oscillates
        opt.step(); sch.step(); a.goal_lr_schedule.step()
        assert abs(a.goal_lr_schedule.get_last_lr()[0] - sch.get_last_lr()[0]) < 1e-15
    assert a.goal_policy_optimizer.lr == a.goal_lr_schedule.get_last_lr()[0]


def test_arena_adam_state_dict_format_roundtrip():
    from porl_amd.agent.por import POR
    a = POR(_args(8, 16, 1), 50, 0.9, 10.0)
    assert a.v_optimizer.state_dict()["state"] == {}
    a._engine.adam_m_vf[:a._engine.n_vf].copy_(torch.arange(a._engine.n_vf, dtype=torch.float32))   # (flat groups carry a zero tail)
    a.v_optimizer.step_count = 7
    sd = a.v_optimizer.state_dict()
    names = [n for n, _ in a.vf.named_parameters()]
    assert sorted(sd["state"].keys()) == list(range(len(names)))
    assert sd["param_groups"][0]["params"] == list(range(len(names)))
    assert sd["param_groups"][0]["betas"] == (0.9, 0.999) and sd["param_groups"][0]["eps"] == 1e-8
    for i, (_, p) in enumerate(a.vf.named_parameters()):
        assert sd["state"][i]["exp_avg"].shape == p.shape and float(sd["state"][i]["step"]) == 7.0
    # a genuine torch.optim.Adam accepts the dict (checkpoint interchange with the reference)
    ref_params = [torch.nn.Parameter(torch.zeros_like(p)) for p in a.vf.parameters()]
    opt = torch.optim.Adam(ref_params, lr=1e-4)
    opt.load_state_dict(sd)
    b = POR(_args(8, 16, 1), 50, 0.9, 10.0)
    b.v_optimizer.load_state_dict(opt.state_dict())
    assert b.v_optimizer.step_count == 7
    ma, _ = a.v_optimizer._moments()
    mb, _ = b.v_optimizer._moments()
    assert all(torch.equal(x, y) for x, y in zip(ma, mb))        # (padding floats are not part of the state)


def test_shard_bounds_partition():
    from porl_amd.parallel import shard_bounds
    for n, w in [(10_000_000, 8), (1000, 3), (7, 8)]:
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


# ---- data-parallel equivalence over gloo, world_size 2 ---------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _dp_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from oracle.por_oracle import PorOracle, value_step, twin_param_names
    from porl_amd.parallel import GradExchange, shard_bounds
    from porl_amd.util.init import build_por_state_dict
    from porl_amd.util.synth import make_rows, split_rows
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ex = GradExchange()
    assert ex.world_size == world and ex.rank == rank
    S, H, L, Bl = 12, 32, 2, 16
    rows = make_rows(world * 64, S, 2, seed=5)
    lo, hi = shard_bounds(rows.shape[0], rank, world)
    local = rows[lo:hi][:Bl]                                   # this rank's sub-batch of its shard
    o = PorOracle(build_por_state_dict(S, H, L, seed=0), S, H, L)
    s, r, sp, d, a = split_rows(local, S, 2)
    inv = 1.0 / (world * Bl)
    v_loss, tv, G = value_step(o.P, "vf", "v_target", o.adam_v, np.ascontiguousarray(s), np.ascontiguousarray(sp),
                               r, d, L, False, 0.9, 0.99, 0.005, inv_batch=inv)
    names = twin_param_names("vf", L, False)
    flat = torch.from_numpy(np.concatenate([G[n].ravel() for n in names]))
    stats = torch.tensor([v_loss, 0.0, float(rank)], dtype=torch.float32)
    ex.allreduce_sum_(flat)
    ex.allreduce_stats_(stats)
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), flat=flat.numpy(), stats=stats.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_dp_gradient_sum_equals_global_batch(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "dp.npz")
    # single-process oracle on the concatenated global minibatch
    from oracle.por_oracle import PorOracle, value_step, twin_param_names
    from porl_amd.parallel import shard_bounds
    from porl_amd.util.init import build_por_state_dict
    from porl_amd.util.synth import make_rows, split_rows
    S, H, L, Bl = 12, 32, 2, 16
    rows = make_rows(world * 64, S, 2, seed=5)
    glob = np.concatenate([rows[slice(*shard_bounds(rows.shape[0], r, world))][:Bl] for r in range(world)])
    o = PorOracle(build_por_state_dict(S, H, L, seed=0), S, H, L)
    s, r, sp, d, a = split_rows(glob, S, 2)
    v_loss, tv, G = value_step(o.P, "vf", "v_target", o.adam_v, np.ascontiguousarray(s), np.ascontiguousarray(sp),
                               r, d, L, False, 0.9, 0.99, 0.005)
    ref = np.concatenate([G[n].ravel() for n in twin_param_names("vf", L, False)])
    np.testing.assert_allclose(got["flat"], ref, atol=1e-7, rtol=1e-5)
    np.testing.assert_allclose(got["stats"][0], v_loss, rtol=1e-6)
    assert got["stats"][2] == 0.0                       # MIN over ranks


def _rs_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from oracle.por_oracle import AdamState, adam_step
    from porl_amd.parallel import GradExchange
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ex = GradExchange()
    n = 4 * world * 25
    rng = np.random.default_rng(3)
    p0 = rng.normal(size=n).astype(np.float32)
    g_local = np.random.default_rng(10 + rank).normal(size=n).astype(np.float32)

    def adam(p, g, st):                                   # the oracle's torch-exact Adam on a flat range
        P = {"w": p}
        adam_step(P, {"w": g}, st)
        return P["w"]

    # (a) all-reduce, Adam on everything
    ga = torch.from_numpy(g_local.copy())
    ex.allreduce_sum_(ga)
    pa = adam(p0.copy(), ga.numpy(), AdamState(1e-3, ["w"]))
    # (b) reduce-scatter, Adam on this rank's slice, all-gather of the parameters
    gb, pb = torch.from_numpy(g_local.copy()), torch.from_numpy(p0.copy())
    assert ex.can_shard(gb)
    gs = torch.empty(n // world)
    ex.reduce_scatter_sum(gb, gs)
    sl = ex.slice_of(pb)
    sl.copy_(torch.from_numpy(adam(sl.numpy().copy(), gs.numpy(), AdamState(1e-3, ["w"]))))
    ex.all_gather_(pb)
    assert np.array_equal(pb.numpy(), pa), "sharded Adam != all-reduce Adam"
    assert not ex.can_shard(torch.zeros(4 * world + 4))       # slices must be equal and 16-byte aligned
    if rank == 0:
        np.save(os.path.join(out_dir, "rs_ok.npy"), pb.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_scatter_sharded_adam_all_gather_equals_all_reduce(tmp_path):
    """SURVEY.md §5.8 exchange (GradExchange.reduce_scatter_sum / slice_of / all_gather_): bit-equal to the all-reduce
    path, world_size 2 over gloo; the device version of the same sequence is tests/test_dp_gpu.py."""
    mp.spawn(_rs_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "rs_ok.npy")


def test_flat_groups_split_evenly_for_every_world_size_up_to_8():
    from porl_amd.agent.por import POR
    torch.manual_seed(0)
    eng = POR(_args(60, 64, 2), 1000, 0.9, 10.0)._engine
    for flat in (eng.params_vf, eng.grads_vf, eng.adam_m_vf, eng.params_tgt, eng.params_pol, eng.grads_pol):
        assert all(flat.numel() % (4 * w) == 0 for w in range(1, 9))
    assert eng.params_vf.numel() >= eng.n_vf and not eng.params_vf[eng.n_vf:].any()


def test_pack_csv_dir_concatenates_the_reference_shard_format(tmp_path):
    """dataloader/dataloader.py:19-20 reads `np.loadtxt(f, delimiter=',').reshape(-1, row_width)` per shard."""
    import numpy as np
    from porl_amd.dataloader import pack_csv_dir
    rng = np.random.default_rng(0)
    width = 12
    shards = [rng.normal(size=(n, width)).astype(np.float32) for n in (100, 100, 37)]
    for i, s in enumerate(shards):
        np.savetxt(tmp_path / f"dataset_{i}.csv", s.reshape(1, -1) if i == 2 else s, delimiter=",", fmt="%.9g")
    out = pack_csv_dir(str(tmp_path), str(tmp_path / "packed.npy"), width)
    rows = np.load(out, mmap_mode="r")
    assert rows.dtype == np.float32 and rows.shape == (237, width)
    assert np.array_equal(np.asarray(rows), np.concatenate(shards))
