"""`torch.ops.porl_hip.*` — the C-ABI entry points (include/porl_hip.h) registered as PyTorch custom operators
(BASELINE north_star: "driven from Python through PyTorch-ROCm custom ops"; SURVEY.md §8(b) names them).

The operators are a thin layer: tensors in, raw pointers + sizes + the current HIP stream of the tensors' device
handed to libporl_hip.so.  Engines (which own flat parameter groups, Adam state and workspace) are referred to by an
integer handle obtained from `register_engine(obj)`; `por_step` / `iql_value_step` / `awr_policy_step` / `cql_step`
are what the agent classes run per minibatch, `replay_gather` / `adam_ema_sweep` / `mlp_forward` / `gemm_f32` /
`sample_indices` / `state2costmap` are the building blocks.

Only the CUDA (= HIP on ROCm) dispatch key has an implementation.  There is NO CPU kernel: calling an operator with
CPU tensors raises `NativeError`, like every other entry of this package.
"""
from __future__ import annotations

import torch

from . import _native as N
from . import engine as E

LIB = torch.library.Library("porl_hip", "DEF")

_engines: dict[int, object] = {}


def register_engine(obj) -> int:
    """Handle for an IqlEngine (POR / SORL agents: `agent._engine`) or a QnetEngine (`trainer._engine`)."""
    h = id(obj)
    _engines[h] = obj
    return h


def release_engine(handle: int) -> None:
    _engines.pop(handle, None)


def _eng(handle):
    try:
        return _engines[handle]
    except KeyError:
        raise RuntimeError(f"porl_hip: unknown engine handle {handle}; use porl_amd.ops.register_engine") from None


SCHEMAS = {
    # one full update of an IQL-family agent on a loaded minibatch (agent/por.py:73-112, agent/sorl.py:78-128)
    "por_step": "(int engine, Tensor obs, Tensor next_obs, Tensor rewards, Tensor terminals, Tensor? pol_target, "
                "float tau, float discount, float alpha, float ema_beta, float value_lr, float policy_lr, "
                "int value_step, int policy_step) -> Tensor",
    # value half (por.py:81-93): forward, expectile loss, backward, Adam + target EMA -> stats[0:1]
    "iql_value_step": "(int engine, Tensor obs, Tensor next_obs, Tensor rewards, Tensor terminals, float tau, "
                      "float discount, float ema_beta, float value_lr, int value_step) -> Tensor",
    # policy half (por.py:97-110) on the minibatch the value half loaded -> stats[1:3]
    # (`like`: any tensor on the engine's device — the dispatcher needs one to pick the HIP kernel)
    "awr_policy_step": "(int engine, Tensor like, float alpha, float policy_lr, int policy_step) -> Tensor",
    # CQL(H) learn() (src/porl/train/cql_trainer.py:88-124) -> [loss, td, penalty]
    "cql_step": "(int engine, Tensor states, Tensor actions, Tensor rewards, Tensor next_states, Tensor dones, "
                "float gamma, float alpha, float lr, int step) -> Tensor",
    "replay_gather": "(Tensor rows, Tensor idx) -> Tensor",
    "adam_ema_sweep": "(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, Tensor(d!)? target, float lr, int step, "
                      "float beta1, float beta2, float eps, float ema_beta) -> ()",
    # forward-only: which = 0 policy mean (B, D); 1 online twins -> (2, B); 2 target twins -> (2, B)
    "mlp_forward": "(int engine, Tensor x, int which) -> Tensor",
    # C = act(A @ B^T + bias): fp32 MFMA GEMM, A (M, K), B (N, K); act 0 none, 1 relu, 2 tanh
    "gemm_f32": "(Tensor a, Tensor b, Tensor? bias, int act) -> Tensor",
    "sample_indices": "(int n_rows, int batch, int seed, int step, Tensor like) -> Tensor",
    "state2costmap": "(Tensor(a!) state, int angle_bins, int dist_bins) -> Tensor",
}
for _name, _schema in SCHEMAS.items():
    LIB.define(_name + _schema)


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and t.device.type != "cuda":
            raise N.NativeError("porl_hip operators run on a HIP device only (tensor on %s); there is no CPU kernel" % t.device)


def _hyper(eng, **kw):
    return eng.hyper(**kw)


def _por_step(engine, obs, next_obs, rewards, terminals, pol_target, tau, discount, alpha, ema_beta, value_lr,
              policy_lr, value_step, policy_step):
    _need_cuda(obs, next_obs, rewards, terminals, pol_target)
    eng = _eng(engine)
    eng.join()
    eng.set_mode(E.IqlEngine.MODE_FOLD_COMBINE)
    B = eng.load_batch(obs, next_obs, rewards, terminals, next_obs if pol_target is None else pol_target)
    eng.step(eng.hyper(tau=tau, discount=discount, alpha=alpha, ema_beta=ema_beta, inv_batch=1.0 / B,
                       value_lr=value_lr, policy_lr=policy_lr, value_step=value_step, policy_step=policy_step))
    return eng.stats[:3]


def _iql_value_step(engine, obs, next_obs, rewards, terminals, tau, discount, ema_beta, value_lr, value_step):
    _need_cuda(obs, next_obs, rewards, terminals)
    eng = _eng(engine)
    eng.join()
    eng.set_mode(E.IqlEngine.MODE_FOLD_COMBINE)
    B = eng.load_batch(obs, next_obs, rewards, terminals, next_obs if eng.cfg.pol_out_dim == eng.cfg.obs_dim else None)
    hp = eng.hyper(tau=tau, discount=discount, ema_beta=ema_beta, inv_batch=1.0 / B, value_lr=value_lr,
                   value_step=value_step)
    eng.value_backward(hp)
    eng.value_apply(hp)
    return eng.stats[:1]


def _awr_policy_step(engine, like, alpha, policy_lr, policy_step):
    _need_cuda(like)
    eng = _eng(engine)
    hp = eng.hyper(alpha=alpha, policy_lr=policy_lr, policy_step=policy_step, inv_batch=1.0 / eng.last_batch)
    eng.policy_backward(hp)
    eng.policy_apply(hp)
    return eng.stats[1:3]


def _cql_step(engine, states, actions, rewards, next_states, dones, gamma, alpha, lr, step):
    _need_cuda(states, actions, rewards, next_states, dones)
    eng = _eng(engine)
    B = eng.load_batch(states, actions, rewards, next_states, dones)
    eng.learn(eng.hyper(gamma, alpha, 1.0 / B, step, lr))
    return eng.stats[:3]


def _replay_gather(rows, idx):
    _need_cuda(rows, idx)
    return E.gather_rows(rows, idx)


def _adam_ema_sweep(p, g, m, v, target, lr, step, beta1, beta2, eps, ema_beta):
    _need_cuda(p, g, m, v, target)
    E.adam_ema(p, g, m, v, target, lr, step, beta1, beta2, eps, ema_beta)


def _mlp_forward(engine, x, which):
    _need_cuda(x)
    eng = _eng(engine)
    if which == 0:
        return eng.forward_policy(x)
    return torch.stack(eng.forward_value(x, target=which == 2))


def _gemm_f32(a, b, bias, act):
    _need_cuda(a, b, bias)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1] or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise RuntimeError("gemm_f32: expected fp32 A (M, K) and B (N, K)")
    a, b = a.contiguous(), b.contiguous()
    M, K = a.shape
    Nn = b.shape[0]
    out = torch.empty(M, Nn, dtype=torch.float32, device=a.device)
    E.gemm_f32(0, a, b, M, Nn, K, K, K, out, Nn, bias=bias, act=act)
    return out


def _sample_indices(n_rows, batch, seed, step, like):
    _need_cuda(like)
    return E.sample_indices(n_rows, batch, seed, step, device=like.device)


def _state2costmap(state, angle_bins, dist_bins):
    from .util.costmap import state2costmap
    return state2costmap(state, angle_bins, dist_bins)


_IMPLS = dict(por_step=_por_step, iql_value_step=_iql_value_step, awr_policy_step=_awr_policy_step, cql_step=_cql_step,
              replay_gather=_replay_gather, adam_ema_sweep=_adam_ema_sweep, mlp_forward=_mlp_forward, gemm_f32=_gemm_f32,
              sample_indices=_sample_indices, state2costmap=_state2costmap)


def _no_cpu(name):
    def impl(*args, **kwargs):
        raise N.NativeError(f"porl_hip::{name} has no CPU kernel (HIP / gfx950 only)")
    return impl


for _name, _fn in _IMPLS.items():
    LIB.impl(_name, _fn, "CUDA")
    LIB.impl(_name, _no_cpu(_name), "CPU")
