#!/usr/bin/env python3
"""Golden-vector generator.  TEST INFRASTRUCTURE — runs ONLY in the build container.

Imports the reference implementation from /root/reference (read-only, CPU, torch 2.10) and records
seeded inputs -> reference outputs as small fixtures under tests/golden/.  The reference never
travels to the GPU box; only these .npz data files (inputs + expected outputs) do.

What is pinned:
  por_*   : POR.por_residual_update        (/root/reference/agent/por.py:73-112)
  sorl_*  : SORL.update / SORL.vf_update   (/root/reference/agent/sorl.py:78-152), backbone=None
  cql_*   : CQLTrainer.learn               (/root/reference/src/porl/train/cql_trainer.py:88-124)
            called as unbound methods on object.__new__(CQLTrainer) — the constructor is broken
            upstream; gymnasium/ipdb/tensorboard are stubbed (not installed).
  replay_*: ReplayBuffer.push/sample index streams under np.random.seed
            (/root/reference/buffer/replay_buffer.py:33-75)
  costmap_*: state2costmap (/root/reference/util/costmap.py:7-64)
  fasternet_*: FasterNet(3, 256).forward_cls (/root/reference/agent/fasternet.py:428-438) in eval mode and in
            train mode (batch-stat BatchNorm, running-stat update, DropPath masks replayed from the seed)
  sorl_enc_*: SORL.update with the FasterNet backbone (/root/reference/agent/sorl.py:78-128)
  per_*   : PrioritizedReplayBuffer.add/sample/update_priorities under random.seed; per_trainer_*: PERTrainer.learn; dqn_* / ddqn_*: DQNTrainer.learn / DDQNTrainer.learn; dddqn_*: DDDQNTrainer.learn on DuelingQNetwork
            (/root/reference/src/porl/train/dqn_per_trainer.py:67-123)
            (/root/reference/src/porl/buffer/prioritized_replay_buffer.py:36-108, sum_tree.py:4-77)

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [dddqn] </dev/null
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "src"))
sys.dont_write_bytecode = True

from porl_amd.util.synth import make_rows, split_rows, make_discrete_transitions  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def sd_np(module_or_sd):
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def adam_np(opt, names):
    """Flatten torch Adam state to {name.exp_avg, name.exp_avg_sq} + step; `names` orders params."""
    out = {}
    params = [p for g in opt.param_groups for p in g["params"]]
    assert len(params) == len(names)
    step = None
    for n, p in zip(names, params):
        st = opt.state[p]
        out[n + ".exp_avg"] = st["exp_avg"].numpy().copy()
        out[n + ".exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
        step = float(st["step"])
    out["__step__"] = np.float64(step)
    return out


def checksum(arr: np.ndarray, salt: int):
    """(sum, abs-sum, 16 sampled values) in float64 — compact pin for tensors too big to store."""
    a = arr.astype(np.float64).ravel()
    idx = np.random.default_rng(1000 + salt).integers(0, a.size, size=16)
    return np.concatenate([[a.sum(), np.abs(a).sum()], a[idx]])


def pack(prefix, d):
    return {prefix + k: v for k, v in d.items()}


# --------------------------------------------------------------------------------------------
def gen_por(name, S, H, L, layer_norm, B, K, full, seed_model=0, seed_data=1, A=2,
            tau=0.9, alpha=10.0, max_steps=1000):
    from agent.por import POR
    args = SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=layer_norm,
                           feature_dim=256, action_size=A)
    torch.manual_seed(seed_model)
    agent = POR(args, max_steps, tau, alpha)
    rows = make_rows(K * B, S, A, seed=seed_data)
    init = sd_np(agent)
    v_losses, g_losses, lrs = [], [], []
    for k in range(K):
        batch = torch.from_numpy(rows[k * B:(k + 1) * B])
        s, r, sp, d, a = split_rows(batch, S, A)
        vl, gl = agent.por_residual_update(s, sp, r, d)
        v_losses.append(vl)
        g_losses.append(gl)
        lrs.append(agent.goal_lr_schedule.get_last_lr()[0])
    final = sd_np(agent)
    meta = dict(S=S, H=H, L=L, layer_norm=int(layer_norm), B=B, K=K, A=A, seed_model=seed_model,
                seed_data=seed_data, tau=tau, alpha=alpha, max_steps=max_steps,
                discount=0.99, beta=0.005, value_lr=1e-4, policy_lr=1e-4)
    out = {"meta_" + k: np.float64(v) for k, v in meta.items()}
    out["v_loss"] = np.array(v_losses, dtype=np.float64)
    out["g_loss"] = np.array(g_losses, dtype=np.float64)
    out["goal_lr_after"] = np.array(lrs, dtype=np.float64)
    keys = list(final.keys())
    out["keys"] = np.array(keys)
    vf_names = [n for n, _ in agent.vf.named_parameters(prefix="vf")]
    gp_names = [n for n, _ in agent.goal_policy.named_parameters(prefix="goal_policy")]
    if full:
        out.update(pack("init/", init))
        out.update(pack("final/", final))
        out.update(pack("adam_v/", adam_np(agent.v_optimizer, vf_names)))
        out.update(pack("adam_g/", adam_np(agent.goal_policy_optimizer, gp_names)))
    else:
        out["init_cks"] = np.stack([checksum(init[k], i) for i, k in enumerate(keys)])
        out["final_cks"] = np.stack([checksum(final[k], i) for i, k in enumerate(keys)])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: v_loss={v_losses} g_loss={g_losses}")


def gen_sorl(name, S, H, L, layer_norm, B, K, A=2, alpha=3.0, tau=0.9, seed_model=0, seed_data=2,
             max_steps=1000, vf_only=False):
    from agent.sorl import SORL
    args = SimpleNamespace(state_size=S, hidden_dim=H, n_hidden=L, layer_norm=layer_norm,
                           feature_dim=256, action_size=A)
    torch.manual_seed(seed_model)
    agent = SORL(args, max_steps, tau, alpha)
    rows = make_rows(K * B, S, A, seed=seed_data)
    init = sd_np(agent)
    v_losses, g_losses = [], []
    for k in range(K):
        batch = torch.from_numpy(rows[k * B:(k + 1) * B])
        s, r, sp, d, a = split_rows(batch, S, A)
        if vf_only:
            v_losses.append(agent.vf_update(s, a, r, sp, d))
        else:
            vl, gl = agent.update(s, a, r, sp, d)
            v_losses.append(vl)
            g_losses.append(gl)
    final = sd_np(agent)
    # select_action on the first 8 observations with the final weights (sorl.py:71-76)
    act = agent.select_action(torch.from_numpy(rows[:8, :S].copy()))
    meta = dict(S=S, H=H, L=L, layer_norm=int(layer_norm), B=B, K=K, A=A, seed_model=seed_model,
                seed_data=seed_data, tau=tau, alpha=alpha, max_steps=max_steps,
                discount=0.99, beta=0.005, value_lr=1e-4, policy_lr=1e-4, vf_only=int(vf_only))
    out = {"meta_" + k: np.float64(v) for k, v in meta.items()}
    out["v_loss"] = np.array(v_losses, dtype=np.float64)
    out["g_loss"] = np.array(g_losses, dtype=np.float64)
    out["select_action"] = act
    out["keys"] = np.array(list(final.keys()))
    out.update(pack("init/", init))
    out.update(pack("final/", final))
    v_names = [n for n, _ in agent.v_net.named_parameters(prefix="v_net")]
    p_names = [n for n, _ in agent.policy.named_parameters(prefix="policy")]
    out.update(pack("adam_v/", adam_np(agent.v_optimizer, v_names)))
    if not vf_only:
        out.update(pack("adam_g/", adam_np(agent.policy_optimizer, p_names)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: v_loss={v_losses} g_loss={g_losses}")


def _stub_cql_imports():
    """gymnasium / ipdb / tensorboard are not installed; porl.train imports them at module scope."""
    for mod in ("gymnasium", "ipdb", "tensorboard", "torch.utils.tensorboard", "matplotlib",
                "matplotlib.pyplot"):
        if mod not in sys.modules:
            try:
                __import__(mod)
            except Exception:
                m = types.ModuleType(mod)
                m.__path__ = []
                sys.modules[mod] = m
    tb = sys.modules["torch.utils.tensorboard"]
    if not hasattr(tb, "SummaryWriter"):
        class SummaryWriter:  # no-op
            def __init__(self, *a, **k): pass
            def add_scalar(self, *a, **k): pass
            def add_hparams(self, *a, **k): pass
            def close(self): pass
        tb.SummaryWriter = SummaryWriter
    sys.modules["ipdb"].set_trace = lambda *a, **k: None
    gym = sys.modules["gymnasium"]
    if not hasattr(gym, "Env"):
        gym.Env = object
        gym.make = lambda *a, **k: None
        gym.spaces = types.ModuleType("gymnasium.spaces")


def gen_cql(name, S, A, B, K, N, seed_model=0, seed_data=3, seed_np=7, sync_every=2, gamma=0.99,
            alpha=1):
    _stub_cql_imports()
    from porl.train.cql_trainer import CQLTrainer
    from porl.net.q_network import QNetwork
    from porl.buffer.replaybuffer import ReplayBuffer
    dev = torch.device("cpu")
    torch.manual_seed(seed_model)
    t = object.__new__(CQLTrainer)
    # mirrors dqn_trainer.py:66-71 (nets, load_state_dict, Adam lr=5e-4)
    t.q_network = QNetwork(S, A).to(dev)
    t.target_network = QNetwork(S, A).to(dev)
    t.target_network.load_state_dict(t.q_network.state_dict())
    t.target_network.eval()
    t.optimizer = torch.optim.Adam(t.q_network.parameters(), lr=0.0005)
    t.replay_buffer = ReplayBuffer(N, (S,), dev)
    t.batch_size, t.gamma, t.alpha, t.action_size, t.device = B, gamma, alpha, A, dev
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=seed_data)
    for i in range(N):
        t.replay_buffer.push(st[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
    init = sd_np(t.q_network)
    np.random.seed(seed_np)
    # record the index stream the reference's sampler will draw, then replay the same seed
    idx = np.stack([np.random.choice(N, B, replace=False) for _ in range(K)])
    np.random.seed(seed_np)
    losses = []
    for k in range(K):
        losses.append(CQLTrainer.learn(t))
        if (k + 1) % sync_every == 0:  # hard target sync, dqn_trainer.py:195-196
            t.target_network.load_state_dict(t.q_network.state_dict())
    # one standalone penalty evaluation (cql_trainer.py:60-86) on the first K-step batch
    pen = float(CQLTrainer.compute_cql_penalty(t, torch.from_numpy(st[idx[0]]),
                                               torch.from_numpy(ac[idx[0]])))
    final = sd_np(t.q_network)
    final_t = sd_np(t.target_network)
    names = [n for n, _ in t.q_network.named_parameters()]
    meta = dict(S=S, A=A, B=B, K=K, N=N, seed_model=seed_model, seed_data=seed_data, seed_np=seed_np,
                sync_every=sync_every, gamma=gamma, alpha=alpha, lr=0.0005)
    out = {"meta_" + k: np.float64(v) for k, v in meta.items()}
    out["loss"] = np.array(losses, dtype=np.float64)
    out["indices"] = idx
    out["penalty_final_on_batch0"] = np.float64(pen)
    out["keys"] = np.array(list(final.keys()))
    out.update(pack("init/", init))
    out.update(pack("final/", final))
    out.update(pack("final_target/", final_t))
    out.update(pack("adam/", adam_np(t.optimizer, names)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={losses} pen={pen}")


def gen_replay(name, N, cap, S, B, K, seed_np=11):
    from buffer.replay_buffer import ReplayBuffer
    rb = ReplayBuffer(cap, (S,), torch.device("cpu"))
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, 4, seed=5)
    for i in range(N):  # N > cap exercises the ring wrap (replay_buffer.py:50-51)
        rb.push(st[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
    np.random.seed(seed_np)
    samples = [rb.sample(B) for _ in range(K)]
    out = dict(meta_N=np.float64(N), meta_cap=np.float64(cap), meta_S=np.float64(S),
               meta_B=np.float64(B), meta_K=np.float64(K), meta_seed_np=np.float64(seed_np),
               size=np.float64(len(rb)), position=np.float64(rb.position))
    for k, (s, a, r, n, d) in enumerate(samples):
        out[f"s{k}"] = s.numpy(); out[f"a{k}"] = a.numpy(); out[f"r{k}"] = r.numpy()
        out[f"n{k}"] = n.numpy(); out[f"d{k}"] = d.numpy()
    out["dtype_names"] = np.array([str(t.dtype) for t in samples[0]])
    # B > size must raise ValueError (numpy), replay_buffer.py:64
    try:
        rb.sample(len(rb) + 1)
        out["oversample_raises"] = np.float64(0)
    except ValueError:
        out["oversample_raises"] = np.float64(1)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: size={len(rb)} pos={rb.position}")


def gen_costmap(name, B=24, seed=13):
    """state2costmap (util/costmap.py:7-64): inputs, the in-place-modified state, and the nonzero pixels."""
    from util.costmap import state2costmap
    rng = np.random.default_rng(seed)
    st = np.empty((B, 362), dtype=np.float32)
    st[:, :360] = rng.uniform(0.15, 3.9, size=(B, 360))
    st[:, 360:] = rng.uniform(-3.0, 3.0, size=(B, 2))
    st[0, 5] = 9.5; st[0, 200] = 8.5; st[1, 17] = 0.0; st[2, 360:] = (0.001, 0.0)      # > 8 -> 0; bin 0; dist bin 0 (wraps)
    st[3, 360:] = (-2.0, 1e-4); st[4, 360:] = (-2.0, -1e-4); st[5, 360:] = (0.0, 3.9)  # angle clamp edges, far goal
    st[6, 360:] = (20.0, 20.0)                                                          # goal coords > 8 are zeroed too
    st[7, :360] = 3.99
    x = torch.from_numpy(st.copy())
    out = state2costmap(x).contiguous().numpy()
    nz = np.argwhere(out != 0).astype(np.int32)
    assert np.all(out[out != 0] == 1.0)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), state_in=st, state_after=x.numpy(), nonzero=nz,
                        shape=np.array(out.shape))
    print(f"{name}: {nz.shape[0]} nonzero pixels")


def fasternet_states(B, seed):
    rng = np.random.default_rng(seed)
    st = np.empty((B, 362), dtype=np.float32)
    st[:, :360] = rng.uniform(0.15, 3.9, size=(B, 360))
    st[:, 360:] = rng.uniform(-3.0, 3.0, size=(B, 2))
    return st


def replay_drop_scale(seed, B, probs, calls):
    """The factors drop_path (fasternet.py:86-93) draws during `calls` forwards after torch.manual_seed(seed):
    one bernoulli_(keep) of shape (B,1,1,1) per block with drop_prob > 0, nothing else touches the generator."""
    torch.manual_seed(seed)
    out = []
    for _ in range(calls):
        rows = []
        for p in probs:
            if p > 0:
                rows.append((torch.empty(B, 1, 1, 1).bernoulli_(1 - p) / (1 - p)).view(B).numpy().copy())
            else:
                rows.append(np.ones(B, dtype=np.float32))
        out.append(np.stack(rows))
    return out


def gen_fasternet(name, B=5, seed_model=0, seed_data=21):
    """Encoder forward: weights are NOT stored (4 MB of noise) — the drop-in's constructor reproduces them from
    seed_model and the fixture pins them by per-tensor checksums; stored are states, features and BatchNorm
    statistics, plus small taps of intermediate activations that pin the oracle stage by stage."""
    from agent.fasternet import FasterNet
    torch.manual_seed(seed_model)
    m = FasterNet(3, 256)
    sd0 = sd_np(m)
    out = {"seed_model": np.int64(seed_model), "B": np.int64(B)}
    for i, (k, v) in enumerate(sd0.items()):
        if v.ndim:
            out["wsum." + k] = checksum(v, i)
    st = fasternet_states(B, seed_data)
    st2 = fasternet_states(B, seed_data + 1)
    out["states"], out["states2"] = st, st2
    probs = [float(x) for x in torch.linspace(0, 0.1, 3)]
    out["drop_probs"] = np.array(probs)

    taps = {}
    hooks = []
    for key, mod in (("patch_embed", m.patch_embed), ("stages.0", m.stages[0]), ("stages.1", m.stages[1]),
                     ("stages.2", m.stages[2]), ("avgpool_pre_head", m.avgpool_pre_head)):
        hooks.append(mod.register_forward_hook(lambda _m, _i, o, key=key: taps.__setitem__(key, o.detach().numpy().copy())))
    m.eval()
    with torch.no_grad():
        out["feat_eval"] = m(torch.from_numpy(st.copy())).numpy().copy()
    for k, v in taps.items():
        v = v.reshape(v.shape[0], v.shape[1], -1) if v.ndim == 4 else v
        out["tap_eval." + k] = np.concatenate([[v.astype(np.float64).sum(), np.abs(v.astype(np.float64)).sum()],
                                               v[:, :6].astype(np.float64).reshape(v.shape[0], -1)[:, :24].ravel()])
    for h in hooks:
        h.remove()

    # train mode: two consecutive forwards (running stats chain, generator keeps advancing)
    m.train()
    seed_fwd = None
    for cand in range(100, 200):
        ds = replay_drop_scale(cand, B, probs, 2)
        if all((d == 0).any() for d in ds):
            seed_fwd = cand
            break
    assert seed_fwd is not None
    out["seed_fwd"] = np.int64(seed_fwd)
    out["drop_scale1"], out["drop_scale2"] = ds
    torch.manual_seed(seed_fwd)
    with torch.no_grad():
        out["feat_train1"] = m(torch.from_numpy(st.copy())).numpy().copy()
        out["feat_train2"] = m(torch.from_numpy(st2.copy())).numpy().copy()
    for k, v in sd_np(m).items():
        if "running_" in k or "num_batches" in k:
            out["stat_after." + k] = v
    # eval again with the updated running statistics
    m.eval()
    with torch.no_grad():
        out["feat_eval_after"] = m(torch.from_numpy(st.copy())).numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: seed_fwd={seed_fwd} |feat_eval|max={np.abs(out['feat_eval']).max():.4g} "
          f"|feat_train1|max={np.abs(out['feat_train1']).max():.4g}")


def gen_sorl_enc(name, B=6, K=3, H=64, L=2, A=2, F=256, seed_model=0, seed_data=31, seed_fwd=77, alpha=3.0, tau=0.9):
    """SORL.update with backbone=FasterNet(3, F) (sorl_train.py:29-33): K joint updates on (B, 362) states."""
    from agent.fasternet import FasterNet
    from agent.sorl import SORL
    torch.manual_seed(seed_model)
    backbone = FasterNet(3, F)
    args = SimpleNamespace(state_size=362, feature_dim=F, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=A)
    agent = SORL(args, max_steps=50, tau=tau, alpha=alpha, backbone=backbone)
    out = {"seed_model": np.int64(seed_model), "seed_fwd": np.int64(seed_fwd), "meta": np.array([B, K, H, L, A, F]),
           "alpha": np.float64(alpha), "tau": np.float64(tau)}
    sd0 = sd_np(agent)
    for i, (k, v) in enumerate(sd0.items()):
        if v.ndim and k.startswith("backbone."):
            out["wsum." + k] = checksum(v, i)
        elif not k.startswith("backbone."):
            out["init." + k] = v
    rng = np.random.default_rng(seed_data)
    probs = [float(x) for x in torch.linspace(0, 0.1, 3)]
    ds = replay_drop_scale(seed_fwd, B, probs, 2 * K)
    torch.manual_seed(seed_fwd)
    losses = []
    for k in range(K):
        s, s2 = fasternet_states(B, seed_data + 10 * k), fasternet_states(B, seed_data + 10 * k + 1)
        a = rng.uniform(-1, 1, size=(B, A)).astype(np.float32)
        r = rng.normal(size=B).astype(np.float32)
        d = (rng.uniform(size=B) < 0.2).astype(np.float32)
        out[f"s{k}"], out[f"s2{k}"], out[f"a{k}"], out[f"r{k}"], out[f"d{k}"] = s, s2, a, r, d
        out[f"drop_s{k}"], out[f"drop_s2{k}"] = ds[2 * k], ds[2 * k + 1]
        vl, gl = agent.update(torch.from_numpy(s.copy()), torch.from_numpy(a), torch.from_numpy(r),
                              torch.from_numpy(s2.copy()), torch.from_numpy(d))
        losses.append((vl, gl))
    out["losses"] = np.array(losses, dtype=np.float64)
    for k, v in sd_np(agent).items():
        if k.startswith("backbone."):
            if "running_" in k or "num_batches" in k:
                out["final." + k] = v
        else:
            out["final." + k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: losses={losses}")


def gen_por_enc(name, B=4, K=2, H=64, L=2, F=256, seed_model=0, seed_data=53, seed_fwd=78, alpha=10.0, tau=0.9):
    """POR.por_residual_update with backbone=FasterNet(3, F) (agent/por.py:46-57,75-79): heads on the features, the goal
    policy regresses the raw 362-wide next state."""
    from agent.fasternet import FasterNet
    from agent.por import POR
    torch.manual_seed(seed_model)
    backbone = FasterNet(3, F)
    args = SimpleNamespace(state_size=362, feature_dim=F, hidden_dim=H, n_hidden=L, layer_norm=False, action_size=2)
    agent = POR(args, max_steps=50, tau=tau, alpha=alpha, backbone=backbone)
    out = {"seed_model": np.int64(seed_model), "seed_fwd": np.int64(seed_fwd), "meta": np.array([B, K, H, L, F]),
           "alpha": np.float64(alpha), "tau": np.float64(tau)}
    for k, v in sd_np(agent).items():
        if not k.startswith("backbone."):
            out["init." + k] = v
    rng = np.random.default_rng(seed_data)
    torch.manual_seed(seed_fwd)
    losses = []
    for k in range(K):
        s, s2 = fasternet_states(B, seed_data + 10 * k), fasternet_states(B, seed_data + 10 * k + 1)
        r = rng.normal(size=B).astype(np.float32)
        d = (rng.uniform(size=B) < 0.2).astype(np.float32)
        out[f"s{k}"], out[f"s2{k}"], out[f"r{k}"], out[f"d{k}"] = s, s2, r, d
        vl, gl = agent.por_residual_update(torch.from_numpy(s.copy()), torch.from_numpy(s2.copy()), torch.from_numpy(r),
                                           torch.from_numpy(d))
        losses.append((vl, gl))
    out["losses"] = np.array(losses, dtype=np.float64)
    for k, v in sd_np(agent).items():
        if not k.startswith("backbone."):
            out["final." + k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: losses={losses}")


def gen_per(name, cap=300, N=450, S=6, B=64, seed=17):
    """PrioritizedReplayBuffer (src/porl/buffer/prioritized_replay_buffer.py:7-108): ring of `cap` filled with N > cap
    adds, two samples under random.seed, a priority write-back with a duplicated index, a third sample."""
    import random
    from porl.buffer.prioritized_replay_buffer import PrioritizedReplayBuffer
    rng = np.random.default_rng(seed)
    buf = PrioritizedReplayBuffer(cap, alpha=0.6, beta_start=0.4, beta_frames=1000)
    st = rng.normal(size=(N, S)).astype(np.float32)
    ns = rng.normal(size=(N, S)).astype(np.float32)
    ac = rng.integers(0, 4, size=N)
    rw = rng.normal(size=N).astype(np.float32)
    dn = (rng.uniform(size=N) < 0.1).astype(np.float32)
    td = np.abs(rng.normal(size=N)) + 0.01
    for i in range(N):
        buf.add(td[i], st[i], int(ac[i]), float(rw[i]), ns[i], float(dn[i]))
    out = dict(meta=np.array([cap, N, S, B, seed]), st=st, ns=ns, ac=ac, rw=rw, dn=dn, td=td)
    random.seed(seed)
    for k in range(2):
        s, a, r, n2, d, w, idxs = buf.sample(B)
        out[f"idx{k}"], out[f"w{k}"], out[f"s{k}"], out[f"a{k}"], out[f"r{k}"] = np.array(idxs), w, s, a, r
    new_td = np.abs(rng.normal(size=B))
    upd_idx = np.array(out["idx1"])
    upd_idx[5] = upd_idx[3]                                   # the same leaf twice: the later value must win
    buf.update_priorities(list(upd_idx), new_td)
    out["upd_idx"], out["upd_td"] = upd_idx, new_td
    s, a, r, n2, d, w, idxs = buf.sample(B)
    out["idx2"], out["w2"], out["s2"] = np.array(idxs), w, s
    out["tree_after"] = buf.tree.tree.copy()
    out["total"] = np.float64(buf.tree.total_priority())
    out["beta"] = np.float64(buf.beta)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: total={out['total']:.6f} beta={out['beta']:.4f}")


def gen_per_trainer(name, S=12, A=5, B=64, K=4, N=400, cap=512, seed_model=2, seed_data=23, seed_rand=5, gamma=0.99):
    """PERTrainer.learn (src/porl/train/dqn_per_trainer.py:67-123) as an unbound method on a hand-built object (the
    constructor needs gymnasium): Double-DQN target, the (B,1)x(B,) weighted loss as written, Adam, priority
    write-back, K steps under random.seed."""
    import random
    _stub_cql_imports()
    from porl.train.dqn_per_trainer import PERTrainer
    from porl.net.q_network import QNetwork
    from porl.buffer.prioritized_replay_buffer import PrioritizedReplayBuffer
    dev = torch.device("cpu")
    torch.manual_seed(seed_model)
    t = object.__new__(PERTrainer)
    t.q_network = QNetwork(S, A).to(dev)
    t.target_network = QNetwork(S, A).to(dev)
    t.target_network.load_state_dict(t.q_network.state_dict())
    with torch.no_grad():                                      # a target that differs from the online net
        for p in t.target_network.parameters():
            p.add_(0.05 * torch.randn_like(p))
    t.optimizer = torch.optim.Adam(t.q_network.parameters(), lr=0.0005)
    t.memory = PrioritizedReplayBuffer(cap, alpha=0.6, beta_start=0.4, beta_frames=1000)
    t.batch_size, t.gamma, t.device = B, gamma, dev
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=seed_data)
    for i in range(N):
        t.memory.add(1.0, st[i], int(ac[i]), float(rw[i]), ns[i], float(dn[i]))
    out = {"meta": np.array([S, A, B, K, N, cap, seed_model, seed_data, seed_rand]), "gamma": np.float64(gamma)}
    out.update(pack("init/", sd_np(t.q_network)))
    out.update(pack("init_target/", sd_np(t.target_network)))
    random.seed(seed_rand)
    losses = []
    for k in range(K):
        losses.append(PERTrainer.learn(t))
    out["loss"] = np.array(losses, dtype=np.float64)
    out.update(pack("final/", sd_np(t.q_network)))
    out["tree_after"] = t.memory.tree.tree.copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={losses}")


def gen_dqn(name, double, S=10, A=6, B=64, K=5, N=500, seed_model=4, seed_data=29, seed_np=3, gamma=0.99):
    """DQNTrainer.learn (src/porl/train/dqn_trainer.py:93-118) / DDQNTrainer.learn (ddqn_trainer.py:58-99) as unbound
    methods on a hand-built object; the numpy index stream is pinned by np.random.seed."""
    _stub_cql_imports()
    from porl.train.dqn_trainer import DQNTrainer
    from porl.train.ddqn_trainer import DDQNTrainer
    from porl.net.q_network import QNetwork
    from porl.buffer.replaybuffer import ReplayBuffer
    cls = DDQNTrainer if double else DQNTrainer
    dev = torch.device("cpu")
    torch.manual_seed(seed_model)
    t = object.__new__(cls)
    t.q_network = QNetwork(S, A).to(dev)
    t.target_network = QNetwork(S, A).to(dev)
    t.target_network.load_state_dict(t.q_network.state_dict())
    with torch.no_grad():
        for p in t.target_network.parameters():
            p.add_(0.05 * torch.randn_like(p))
    t.optimizer = torch.optim.Adam(t.q_network.parameters(), lr=0.0005)
    t.replay_buffer = ReplayBuffer(N, (S,), dev)
    t.batch_size, t.gamma, t.device = B, gamma, dev
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=seed_data)
    for i in range(N):
        t.replay_buffer.push(st[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
    out = {"meta": np.array([S, A, B, K, N, seed_model, seed_data, seed_np, int(double)]), "gamma": np.float64(gamma)}
    out.update(pack("init/", sd_np(t.q_network)))
    out.update(pack("init_target/", sd_np(t.target_network)))
    np.random.seed(seed_np)
    losses = [cls.learn(t) for _ in range(K)]
    out["loss"] = np.array(losses, dtype=np.float64)
    out.update(pack("final/", sd_np(t.q_network)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={losses}")


def gen_dddqn(name, S=10, A=6, B=64, K=5, N=500, seed_model=8, seed_data=37, seed_np=5, gamma=0.99):
    """DDDQNTrainer.learn (src/porl/train/dddqn_trainer.py:59-103: Double-DQN target) on DuelingQNetwork pairs
    (src/porl/net/q_network.py:33-68: value + advantage streams on 64 features, q = v + a - mean(a)), as an unbound method
    on a hand-built object (the constructor needs gymnasium); numpy's index stream pinned by np.random.seed."""
    _stub_cql_imports()
    from porl.train.dddqn_trainer import DDDQNTrainer
    from porl.net.q_network import DuelingQNetwork
    from porl.buffer.replaybuffer import ReplayBuffer
    dev = torch.device("cpu")
    torch.manual_seed(seed_model)
    t = object.__new__(DDDQNTrainer)
    t.q_network = DuelingQNetwork(S, A).to(dev)
    t.target_network = DuelingQNetwork(S, A).to(dev)
    t.target_network.load_state_dict(t.q_network.state_dict())
    with torch.no_grad():
        for p in t.target_network.parameters():
            p.add_(0.05 * torch.randn_like(p))
    t.optimizer = torch.optim.Adam(t.q_network.parameters(), lr=0.0005)
    t.replay_buffer = ReplayBuffer(N, (S,), dev)
    t.batch_size, t.gamma, t.device = B, gamma, dev
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=seed_data)
    for i in range(N):
        t.replay_buffer.push(st[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
    out = {"meta": np.array([S, A, B, K, N, seed_model, seed_data, seed_np]), "gamma": np.float64(gamma)}
    out.update(pack("init/", sd_np(t.q_network)))
    out.update(pack("init_target/", sd_np(t.target_network)))
    probe = torch.from_numpy(st[:4])
    out["probe_x"] = st[:4]
    out["probe_q0"] = t.q_network(probe).detach().numpy()
    np.random.seed(seed_np)
    losses = [DDDQNTrainer.learn(t) for _ in range(K)]
    out["loss"] = np.array(losses, dtype=np.float64)
    out.update(pack("final/", sd_np(t.q_network)))
    out["probe_q"] = t.q_network(probe).detach().numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={losses}")


def gen_bcq(name, S=10, A=6, B=64, K=5, KP=6, N=500, seed_model=5, seed_data=31, seed_np=11, gamma=0.99, threshold=0.17):
    """bcq_behavior_pretrain (src/porl/policy/bcq.py:23-47) for KP epochs, then bcq_learn (:50-86) for K steps, on a
    hand-built BCQTrainer (its constructor needs gymnasium); numpy's index stream pinned by np.random.seed.  The
    threshold sits inside the spread of the pre-trained behaviour probabilities so the mask has both values."""
    _stub_cql_imports()
    from porl.train.bcq_trainer import BCQTrainer
    from porl.policy.bcq import bcq_learn, bcq_behavior_pretrain
    from porl.net.q_network import QNetwork
    from porl.net.behavior_policy import BehaviorPolicy
    from porl.buffer.replaybuffer import ReplayBuffer
    import contextlib, io
    dev = torch.device("cpu")
    torch.manual_seed(seed_model)
    t = object.__new__(BCQTrainer)
    t.q_network = QNetwork(S, A).to(dev)
    t.target_network = QNetwork(S, A).to(dev)
    t.target_network.load_state_dict(t.q_network.state_dict())
    with torch.no_grad():
        for p in t.target_network.parameters():
            p.add_(0.05 * torch.randn_like(p))
    t.optimizer = torch.optim.Adam(t.q_network.parameters(), lr=0.0005)
    t.behavior_policy = BehaviorPolicy(S, A).to(dev)                      # bcq_trainer.py:59-62
    t.behavior_optimizer = torch.optim.Adam(t.behavior_policy.parameters(), lr=0.0005)
    t.replay_buffer = ReplayBuffer(N, (S,), dev)
    t.batch_size, t.gamma, t.device, t.num_epochs, t.threshold = B, gamma, dev, KP, threshold
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=seed_data)
    for i in range(N):
        t.replay_buffer.push(st[i], int(ac[i]), float(rw[i]), ns[i], bool(dn[i]))
    out = {"meta": np.array([S, A, B, K, N, seed_model, seed_data, seed_np, KP]), "gamma": np.float64(gamma),
           "threshold": np.float64(threshold)}
    out.update(pack("init/", sd_np(t.q_network)))
    out.update(pack("init_target/", sd_np(t.target_network)))
    out.update(pack("init_behavior/", sd_np(t.behavior_policy)))
    np.random.seed(seed_np)
    # per-epoch cross-entropy of the pre-training (the reference only prints every 10th): re-run its loop body
    ce = []
    for epoch in range(KP):
        states, actions, _, _, _ = t.replay_buffer.sample(B)
        logits = t.behavior_policy.network(states)
        loss = torch.nn.functional.cross_entropy(logits, actions)
        t.behavior_optimizer.zero_grad(); loss.backward(); t.behavior_optimizer.step()
        ce.append(loss.item())
    # ... and check that this IS what the reference function does, from the same start
    torch.manual_seed(seed_model)
    chk = BehaviorPolicy(S, A)
    chk.load_state_dict({k: torch.from_numpy(v) for k, v in sub_dict(out, "init_behavior/").items()})
    t2 = object.__new__(BCQTrainer)
    t2.behavior_policy, t2.behavior_optimizer = chk, torch.optim.Adam(chk.parameters(), lr=0.0005)
    t2.replay_buffer, t2.batch_size, t2.num_epochs = t.replay_buffer, B, KP
    np.random.seed(seed_np)
    with contextlib.redirect_stdout(io.StringIO()):
        bcq_behavior_pretrain(t2)
    for (k, a), (_, b) in zip(sd_np(chk).items(), sd_np(t.behavior_policy).items()):
        assert np.array_equal(a, b), k
    out["ce_loss"] = np.array(ce, dtype=np.float64)
    out.update(pack("behavior_after/", sd_np(t.behavior_policy)))
    # the mask of the first learn batch (peek with a copy of the RNG state)
    state = np.random.get_state()
    _, _, _, nxt, _ = t.replay_buffer.sample(B)
    np.random.set_state(state)
    with torch.no_grad():
        out["mask0"] = t.behavior_policy.sample(nxt, threshold).numpy()
        out["probs0"] = t.behavior_policy(nxt).numpy()
    losses = [bcq_learn(t) for _ in range(K)]
    out["loss"] = np.array(losses, dtype=np.float64)
    out.update(pack("final/", sd_np(t.q_network)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: ce={ce} loss={losses} mask ones={out['mask0'].mean():.3f}")


def _fill_replay(t, N, S, A, seed_data, reward_scale=1.0):
    from porl.buffer.replaybuffer import ReplayBuffer
    t.replay_buffer = ReplayBuffer(N, (S,), torch.device("cpu"))
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=seed_data)
    for i in range(N):
        t.replay_buffer.push(st[i], int(ac[i]), float(rw[i]) * reward_scale, ns[i], bool(dn[i]))


def gen_qr(name, S=9, A=5, NQ=12, hidden=(48, 40), B=48, K=5, N=400, seed_model=6, seed_data=37, seed_np=13, gamma=0.97,
           kappa=0.6):
    """QRDQNTrainer.learn (src/porl/train/qr_dqn_trainer.py:97-215) on a hand-built trainer (the constructor needs
    gymnasium); kappa below 1 so both Huber branches occur; the target net differs from the online net."""
    _stub_cql_imports()
    from porl.train.qr_dqn_trainer import QRDQNTrainer
    from porl.net.qr_dqn_network import QRNetwork
    dev = torch.device("cpu")
    torch.manual_seed(seed_model)
    t = object.__new__(QRDQNTrainer)
    t.q_network = QRNetwork(S, A, NQ, list(hidden))
    t.target_network = QRNetwork(S, A, NQ, list(hidden))
    t.target_network.load_state_dict(t.q_network.state_dict())
    with torch.no_grad():
        for p in t.target_network.parameters():
            p.add_(0.1 * torch.randn_like(p))
    t.optimizer = torch.optim.Adam(t.q_network.parameters(), lr=5e-4)
    t.batch_size, t.gamma, t.device, t.num_quantiles, t.kappa = B, gamma, dev, NQ, kappa
    i = torch.arange(0, NQ, dtype=torch.float32)
    t.tau = ((2 * i + 1) / (2 * NQ)).unsqueeze(0)
    _fill_replay(t, N, S, A, seed_data)
    out = {"meta": np.array([S, A, NQ, B, K, N, seed_model, seed_data, seed_np] + list(hidden)), "gamma": np.float64(gamma),
           "kappa": np.float64(kappa)}
    out.update(pack("init/", sd_np(t.q_network)))
    out.update(pack("init_target/", sd_np(t.target_network)))
    np.random.seed(seed_np)
    losses = [QRDQNTrainer.learn(t) for _ in range(K)]
    out["loss"] = np.array(losses, dtype=np.float64)
    out.update(pack("final/", sd_np(t.q_network)))
    x = torch.from_numpy(make_discrete_transitions(4, S, A, seed=1)[0])
    with torch.no_grad():
        out["probe_x"] = x.numpy()
        out["probe_mean_q"] = t.q_network.get_mean_q_values(x).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={losses}")


def gen_c51(name, S=9, A=5, NA=21, hidden=(48, 40), B=48, K=5, N=400, seed_model=7, seed_data=41, seed_np=17, gamma=0.97,
            v_min=-4.0, v_max=4.0):
    """C51Trainer.learn (src/porl/train/c51_trainer.py:52-174) on a hand-built trainer; rewards scaled by 2 so that
    projected atoms hit both clamps of the support."""
    _stub_cql_imports()
    from porl.train.c51_trainer import C51Trainer
    from porl.net.categorical_q_network import CategoricalQNetwork
    dev = torch.device("cpu")
    torch.manual_seed(seed_model)
    t = object.__new__(C51Trainer)
    t.q_network = CategoricalQNetwork(S, A, NA, v_min, v_max, hidden_sizes=list(hidden))
    t.target_network = CategoricalQNetwork(S, A, NA, v_min, v_max, hidden_sizes=list(hidden))
    t.target_network.load_state_dict(t.q_network.state_dict())
    with torch.no_grad():
        for p in t.target_network.parameters():
            p.add_(0.1 * torch.randn_like(p))
    t.optimizer = torch.optim.Adam(t.q_network.parameters(), lr=5e-4)
    t.batch_size, t.gamma, t.device = B, gamma, dev
    t.atom_size, t.v_min, t.v_max = NA, v_min, v_max
    t.delta_z = (v_max - v_min) / (NA - 1)
    t.support = torch.linspace(v_min, v_max, NA)
    _fill_replay(t, N, S, A, seed_data, reward_scale=2.0)
    out = {"meta": np.array([S, A, NA, B, K, N, seed_model, seed_data, seed_np] + list(hidden)), "gamma": np.float64(gamma),
           "v_min": np.float64(v_min), "v_max": np.float64(v_max)}
    out.update(pack("init/", sd_np(t.q_network)))
    out.update(pack("init_target/", sd_np(t.target_network)))
    np.random.seed(seed_np)
    losses = [C51Trainer.learn(t) for _ in range(K)]
    out["loss"] = np.array(losses, dtype=np.float64)
    out.update(pack("final/", sd_np(t.q_network)))
    x = torch.from_numpy(make_discrete_transitions(4, S, A, seed=1)[0])
    with torch.no_grad():
        out["probe_x"] = x.numpy()
        out["probe_logp"] = t.q_network(x).numpy()
        out["probe_q"] = t.q_network.get_q_values(x).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={losses}")


def gen_iqn_loss(name, B=37, NP=8, NPP=11, kappa=0.7, seed=19):
    """IQNTrainer.quantile_huber_loss (src/porl/train/iqn_trainer.py:136-149) as an unbound method — the only piece of
    upstream's IQN trainer that runs (its learn() needs a network class that does not exist in the tree) — with the
    gradient w.r.t. the current quantiles from autograd."""
    _stub_cql_imports()
    from porl.train.iqn_trainer import IQNTrainer
    g = torch.Generator().manual_seed(seed)
    cur = torch.randn(B, NP, generator=g).requires_grad_(True)
    tgt = 1.5 * torch.randn(B, NPP, generator=g)
    taus = torch.rand(B, NP, generator=g)
    t = object.__new__(IQNTrainer)
    t.kappa = kappa
    td = tgt.unsqueeze(1) - cur.unsqueeze(2)                                  # iqn_trainer.py:127
    loss = IQNTrainer.quantile_huber_loss(t, td, taus)
    loss.backward()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), cur=cur.detach().numpy(), target=tgt.numpy(), taus=taus.numpy(),
                        kappa=np.float64(kappa), loss=np.float64(loss.item()), dcur=cur.grad.numpy())
    print(f"{name}: loss={loss.item()}")


def gen_iqn(name, S=9, A=5, E=16, H=64, B=48, K=4, NP=8, NPP=6, seed_model=9, seed_data=43, seed_rand=23, gamma=0.97,
            kappa=0.6, lr=5e-4):
    """IQNNetwork.forward (src/porl/net/iqn_network.py:35-72) on the LIVE class of that file, and IQNTrainer.learn
    (src/porl/train/iqn_trainer.py:92-134) — upstream's own method, unmodified — on a hand-built trainer: the constructor
    cannot run (it passes five arguments to a four-argument network class, :58-72), and learn calls `get_q_values`, which the
    live class lacks, so each network instance gets `get_q_values` bound to its own forward (the one reading under which
    lines 92-134 type-check: (B, N, action) quantile values).  Minibatches are fixed (a stub replay buffer hands them out in
    order); the fractions are torch.rand draws after torch.manual_seed(seed_rand + k), regenerated here for the fixture.
    Rewards are scaled by 3 so both Huber branches occur; the target net differs from the online net."""
    _stub_cql_imports()
    from porl.train.iqn_trainer import IQNTrainer
    from porl.net.iqn_network import IQNNetwork
    dev = torch.device("cpu")
    torch.manual_seed(seed_model)
    t = object.__new__(IQNTrainer)
    t.q_network = IQNNetwork(S, A, E, H)
    t.target_network = IQNNetwork(S, A, E, H)
    t.target_network.load_state_dict(t.q_network.state_dict())
    with torch.no_grad():
        for p in t.target_network.parameters():
            p.add_(0.1 * torch.randn_like(p))
    for net in (t.q_network, t.target_network):
        net.get_q_values = net.forward
    t.optimizer = torch.optim.Adam(t.q_network.parameters(), lr=lr)
    t.batch_size, t.gamma, t.device, t.kappa = B, gamma, dev, kappa
    t.num_quantiles_n_prime_loss, t.num_quantiles_n_double_prime_loss = NP, NPP
    st, ac, rw, ns, dn = make_discrete_transitions(B * K, S, A, seed=seed_data)
    rw = (3.0 * rw).astype(np.float32)

    class _Replay:
        k = 0
        def sample(self, batch_size):
            i = slice(self.k * batch_size, (self.k + 1) * batch_size)
            self.k += 1
            return (torch.from_numpy(st[i]), torch.from_numpy(ac[i]).long(), torch.from_numpy(rw[i]),
                    torch.from_numpy(ns[i]), torch.from_numpy(dn[i].astype(np.float32)))
    t.replay_buffer = _Replay()
    out = {"meta": np.array([S, A, E, H, B, K, NP, NPP, seed_model, seed_data, seed_rand]), "gamma": np.float64(gamma),
           "kappa": np.float64(kappa), "lr": np.float64(lr), "states": st, "actions": ac.astype(np.int64), "rewards": rw,
           "next_states": ns, "dones": dn.astype(np.float32)}
    out.update(pack("init/", sd_np(t.q_network)))
    out.update(pack("init_target/", sd_np(t.target_network)))
    # forward golden on the initial network
    g = torch.Generator().manual_seed(seed_rand - 1)
    probe_taus = torch.rand(7, 5, generator=g)
    with torch.no_grad():
        out["probe_x"] = st[:7]
        out["probe_taus"] = probe_taus.numpy()
        out["probe_z"] = t.q_network(torch.from_numpy(st[:7]), probe_taus).numpy()
        out["probe_embed"] = t.q_network.get_quantile_embedding(probe_taus).numpy()
    losses, tp, tpp = [], [], []
    for k in range(K):
        torch.manual_seed(seed_rand + k)
        tp.append(torch.rand(B, NP).numpy())
        tpp.append(torch.rand(B, NPP).numpy())
        torch.manual_seed(seed_rand + k)
        losses.append(IQNTrainer.learn(t))
    out["loss"] = np.array(losses, dtype=np.float64)
    out["taus_prime"], out["taus_double_prime"] = np.stack(tp), np.stack(tpp)
    out.update(pack("final/", sd_np(t.q_network)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss={losses}")


def sub_dict(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1:                      # `gen_golden.py dddqn`: (re)generate one fixture family only
        {"dddqn": lambda: gen_dddqn("dddqn_s10_a6"), "iqn": lambda: gen_iqn("iqn_s9_a5")}[sys.argv[1]]()
        return
    # POR — small, fully stored
    gen_por("por_s60_h64_b32", S=60, H=64, L=2, layer_norm=False, B=32, K=5, full=True)
    gen_por("por_s60_h64_b32_ln", S=60, H=64, L=2, layer_norm=True, B=32, K=5, full=True)
    gen_por("por_s17_h48_l3_b50", S=17, H=48, L=3, layer_norm=False, B=50, K=4, full=True,
            seed_model=3, seed_data=4)                       # ragged dims, 3 hidden layers
    gen_por("por_s60_h256_b256", S=60, H=256, L=2, layer_norm=False, B=256, K=3, full=False)
    # BASELINE configs 1/2 — checksums only (5.6 M params)
    gen_por("por_s60_h1024_b256", S=60, H=1024, L=2, layer_norm=False, B=256, K=3, full=False)
    gen_por("por_s60_h1024_b1024", S=60, H=1024, L=2, layer_norm=False, B=1024, K=3, full=False)
    gen_por("por_s60_h1024_b1024_ln", S=60, H=1024, L=2, layer_norm=True, B=1024, K=2, full=False)
    # SORL (backbone=None): alpha MULTIPLIES (sorl.py:104)
    gen_sorl("sorl_s60_h64_b32", S=60, H=64, L=2, layer_norm=False, B=32, K=5, alpha=3.0)
    gen_sorl("sorl_s362_h64_b16_a10", S=362, H=64, L=2, layer_norm=False, B=16, K=3, alpha=10.0)
    gen_sorl("sorl_vf_s60_h64_b32", S=60, H=64, L=2, layer_norm=False, B=32, K=3, vf_only=True)
    # CQL
    gen_cql("cql_s60_a10_b64", S=60, A=10, B=64, K=6, N=2000)
    gen_cql("cql_s8_a4_b256", S=8, A=4, B=256, K=4, N=1000, seed_model=1, seed_np=9)
    # Replay buffer
    gen_replay("replay_ring", N=700, cap=512, S=8, B=64, K=3)
    # costmap rasteriser
    gen_costmap("costmap_b24")
    # FasterNet costmap encoder (config 5) and SORL with it as backbone
    gen_fasternet("fasternet_b5", B=5)
    gen_sorl_enc("sorl_enc_b6", B=6, K=3)
    gen_por_enc("por_enc_b4", B=4, K=2)
    # prioritized replay (next row)
    gen_per("per_cap300")
    gen_per_trainer("per_trainer_s12_a5")
    gen_dqn("dqn_s10_a6", double=False)
    gen_dqn("ddqn_s10_a6", double=True)
    gen_bcq("bcq_s10_a6")
    gen_dddqn("dddqn_s10_a6")
    gen_qr("qrdqn_s9_a5_n12")
    gen_c51("c51_s9_a5_n21")
    gen_iqn_loss("iqn_quantile_huber")
    gen_iqn("iqn_s9_a5")


if __name__ == "__main__":
    main()
