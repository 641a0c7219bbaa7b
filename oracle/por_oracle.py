"""CPU oracle — a numpy fp32 restatement of the reference's batched update step.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product path (porl_amd/) never does and fails loudly without its HIP library.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py) against golden
vectors produced by running the reference itself in the build container (oracle/gen_golden.py ->
tests/golden/*.npz); the reference ships no hot-path tests of its own.

No autograd: forward, hand-derived backward, Adam, EMA and the cosine schedule are written out
explicitly so that this file is also the arithmetic spec of the HIP kernels.  All arrays are
np.float32; matmuls go through numpy's BLAS (different summation order from both MKL and MFMA, so
comparisons are tolerance-based, never bitwise).

State layout: a dict {reference state_dict key -> np.ndarray}, e.g. 'vf.v1.0.weight'.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

F32 = np.float32       # working precision; set_precision(np.float64) gives the exact-arithmetic "truth"
LOG_STD_MIN, LOG_STD_MAX = -5.0, 2.0      # /root/reference/agent/policy.py:8-9
EXP_ADV_MAX = 100.0                        # /root/reference/agent/por.py:12
LN_EPS = 1e-5                              # torch.nn.LayerNorm default (util/util.py:36-37)


def set_precision(dtype):
    """Switch the oracle's working precision (np.float32 default).  With np.float64 the same formulas
    give the exact-arithmetic result up to 1e-16, which tests use as the common yardstick: two fp32
    implementations (MKL, BLAS, MFMA) agree with it to rounding, but can disagree with EACH OTHER by
    ~1e-5 on a handful of weights whenever a ReLU pre-activation or an Adam denominator sits within
    rounding distance of its kink ("what 1e-5 parity means")."""
    global F32
    F32 = dtype


# ---------------------------------------------------------------------------------------------
# mlp (/root/reference/util/util.py:29-47): Linear [+LayerNorm] + ReLU ... Linear [+act] [+Squeeze]
# ---------------------------------------------------------------------------------------------
def mlp_layer_keys(prefix: str, n_hidden: int, layer_norm: bool):
    """nn.Sequential indices of the reference's mlp(): returns ([(lin_idx, ln_idx|None)...], final_idx)."""
    stride = 3 if layer_norm else 2
    hidden = [(f"{prefix}.{i * stride}", f"{prefix}.{i * stride + 1}" if layer_norm else None)
              for i in range(n_hidden)]
    return hidden, f"{prefix}.{n_hidden * stride}"


def mlp_param_names(prefix: str, n_hidden: int, layer_norm: bool):
    """Parameter names in nn.Module.named_parameters() order."""
    hidden, final = mlp_layer_keys(prefix, n_hidden, layer_norm)
    names = []
    for lin, ln in hidden:
        names += [lin + ".weight", lin + ".bias"]
        if ln is not None:
            names += [ln + ".weight", ln + ".bias"]
    names += [final + ".weight", final + ".bias"]
    return names


def mlp_forward(P, prefix, x, n_hidden, layer_norm=False, out_act=None, keep=True):
    """Returns (out, cache).  out is (B, out_dim).  util/util.py:34-43."""
    hidden, final = mlp_layer_keys(prefix, n_hidden, layer_norm)
    cache = {"x": [], "h": [], "xhat": [], "rstd": []}
    h = x.astype(F32, copy=False)
    for lin, ln in hidden:
        cache["x"].append(h)
        z = h @ P[lin + ".weight"].T + P[lin + ".bias"]
        if ln is not None:
            mu = z.mean(axis=1, keepdims=True, dtype=F32)
            var = ((z - mu) ** 2).mean(axis=1, keepdims=True, dtype=F32)      # biased variance
            rstd = (F32(1.0) / np.sqrt(var + F32(LN_EPS))).astype(F32)
            xhat = (z - mu) * rstd
            z = xhat * P[ln + ".weight"] + P[ln + ".bias"]
            cache["xhat"].append(xhat)
            cache["rstd"].append(rstd)
        h = np.maximum(z, F32(0.0))
        cache["h"].append(h)
    cache["x"].append(h)
    out = h @ P[final + ".weight"].T + P[final + ".bias"]
    if out_act == "tanh":
        out = np.tanh(out).astype(F32)
    cache["out"] = out
    return out.astype(F32, copy=False), cache


def mlp_backward(P, prefix, cache, d_out, n_hidden, layer_norm=False, out_act=None):
    """Gradients of all parameters of one mlp given dL/d(out) of shape (B, out_dim)."""
    hidden, final = mlp_layer_keys(prefix, n_hidden, layer_norm)
    G = {}
    d = d_out.astype(F32, copy=False)
    if out_act == "tanh":
        d = d * (F32(1.0) - cache["out"] ** 2)
    G[final + ".weight"] = d.T @ cache["x"][n_hidden]
    G[final + ".bias"] = d.sum(axis=0, dtype=F32)
    dh = d @ P[final + ".weight"]
    for i in reversed(range(n_hidden)):
        lin, ln = hidden[i]
        dy = dh * (cache["h"][i] > 0)
        if ln is not None:
            xhat, rstd = cache["xhat"][i], cache["rstd"][i]
            G[ln + ".weight"] = (dy * xhat).sum(axis=0, dtype=F32)
            G[ln + ".bias"] = dy.sum(axis=0, dtype=F32)
            dxh = dy * P[ln + ".weight"]
            m1 = dxh.mean(axis=1, keepdims=True, dtype=F32)
            m2 = (dxh * xhat).mean(axis=1, keepdims=True, dtype=F32)
            dz = (dxh - m1 - xhat * m2) * rstd
        else:
            dz = dy
        dz = dz.astype(F32, copy=False)
        G[lin + ".weight"] = dz.T @ cache["x"][i]
        G[lin + ".bias"] = dz.sum(axis=0, dtype=F32)
        if i > 0:
            dh = dz @ P[lin + ".weight"]
    return G


# ---------------------------------------------------------------------------------------------
# torch.optim.Adam single-tensor path, EMA (util/util.py:54-56)
# ---------------------------------------------------------------------------------------------
@dataclass
class AdamState:
    lr: float
    names: list
    m: dict = field(default_factory=dict)
    v: dict = field(default_factory=dict)
    step: int = 0
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8


def adam_scalars(lr, step, beta1=0.9, beta2=0.999):
    """Host-side doubles exactly as torch computes them; returned as python floats."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    return lr / bc1, math.sqrt(bc2)


def adam_step(P, G, st: AdamState, lr=None):
    st.step += 1
    lr = st.lr if lr is None else lr
    step_size, bc2_sqrt = adam_scalars(lr, st.step, st.beta1, st.beta2)
    for n in st.names:
        g = G[n].astype(F32, copy=False)
        if n not in st.m:
            st.m[n] = np.zeros_like(P[n])
            st.v[n] = np.zeros_like(P[n])
        m, v = st.m[n], st.v[n]
        m += F32(1.0 - st.beta1) * (g - m)                       # exp_avg.lerp_(grad, 1-beta1)
        v *= F32(st.beta2)
        v += F32(1.0 - st.beta2) * g * g                          # addcmul_
        denom = np.sqrt(v) / F32(bc2_sqrt) + F32(st.eps)
        P[n] = (P[n] - F32(step_size) * (m / denom)).astype(F32)  # addcdiv_(value=-step_size)


def ema_update(P, tgt_prefix, src_prefix, names_src, beta):
    """target.mul_(1-beta).add_(source, alpha=beta)  (util/util.py:54-56)."""
    for n in names_src:
        t = tgt_prefix + n[len(src_prefix):]
        P[t] = (P[t] * F32(1.0 - beta) + F32(beta) * P[n]).astype(F32)


def cosine_lr(lr0, t, t_max):
    """CosineAnnealingLR(T_max, eta_min=0) closed form; t = number of scheduler.step() calls so far
    (equals torch's recursive form to 2e-19)."""
    return lr0 * (1.0 + math.cos(math.pi * t / t_max)) / 2.0


# ---------------------------------------------------------------------------------------------
# shared IQL-style value step (agent/por.py:77-93 == agent/sorl.py:86-98)
# ---------------------------------------------------------------------------------------------
def asymmetric_l2(u, tau):
    """agent/por.py:15-17 — returns (loss, dloss/du) with mean reduction."""
    w = np.abs(F32(tau) - (u < 0).astype(F32))
    loss = np.mean(w * u * u, dtype=F32)
    dldu = (F32(2.0) * w * u / F32(u.shape[0])).astype(F32)
    return loss, dldu


def twin_forward(P, prefix, x, n_hidden, layer_norm):
    v1, c1 = mlp_forward(P, prefix + ".v1", x, n_hidden, layer_norm)
    v2, c2 = mlp_forward(P, prefix + ".v2", x, n_hidden, layer_norm)
    return v1[:, 0], v2[:, 0], c1, c2


def twin_param_names(prefix, n_hidden, layer_norm):
    return (mlp_param_names(prefix + ".v1", n_hidden, layer_norm)
            + mlp_param_names(prefix + ".v2", n_hidden, layer_norm))


def value_step(P, vf, vt, adam_v: AdamState, obs, next_obs, rew, term, n_hidden, layer_norm,
               tau, discount, beta, inv_batch=None):
    """Returns (v_loss, target_v).  inv_batch lets a data-parallel shard scale by 1/B_global."""
    t1, t2, _, _ = twin_forward(P, vt, next_obs, n_hidden, layer_norm)
    next_v = np.minimum(t1, t2)                                           # value_functions.py:41-42
    target_v = (rew + (F32(1.0) - term) * F32(discount) * next_v).astype(F32)   # por.py:85
    v1, v2, c1, c2 = twin_forward(P, vf, obs, n_hidden, layer_norm)
    G = {}
    v_loss = F32(0.0)
    for v, c, name in ((v1, c1, vf + ".v1"), (v2, c2, vf + ".v2")):
        u = target_v - v
        loss, dldu = asymmetric_l2(u, tau)
        if inv_batch is not None:
            w = np.abs(F32(tau) - (u < 0).astype(F32))
            loss = F32(np.sum(w * u * u, dtype=F32) * F32(inv_batch))
            dldu = (F32(2.0) * w * u * F32(inv_batch)).astype(F32)
        v_loss += loss / F32(2.0)
        dv = (-dldu / F32(2.0))[:, None]                                   # d(loss/2)/dv = -dl/du / 2
        G.update(mlp_backward(P, name, c, dv, n_hidden, layer_norm))
    return float(v_loss), target_v, G


def gaussian_nll(mean, x, log_std):
    """-MultivariateNormal(mean, scale_tril=diag(exp(clamp(log_std)))).log_prob(x)
    (agent/policy.py:18-23).  Returns (nlp (B,), z (B,D), sigma (D,))."""
    D = mean.shape[1]
    ls = np.clip(log_std, F32(LOG_STD_MIN), F32(LOG_STD_MAX)).astype(F32)
    sigma = np.exp(ls).astype(F32)
    z = ((x - mean) / sigma).astype(F32)
    half_log_det = np.sum(np.log(sigma), dtype=F32)
    nlp = F32(0.5) * (F32(D * math.log(2.0 * math.pi)) + np.sum(z * z, axis=1, dtype=F32)) + half_log_det
    return nlp.astype(F32), z, sigma


def policy_nll_grads(z, sigma, log_std, weight_over_b):
    """d/dmean and d/dlog_std of sum_b w_b * nlp_b  (SURVEY.md §8 a2.8)."""
    w = weight_over_b[:, None]
    d_mean = (-(w * z) / sigma).astype(F32)
    inside = ((log_std >= F32(LOG_STD_MIN)) & (log_std <= F32(LOG_STD_MAX))).astype(F32)
    d_log_std = (np.sum(w * (F32(1.0) - z * z), axis=0, dtype=F32) * inside).astype(F32)
    return d_mean, d_log_std


# ---------------------------------------------------------------------------------------------
# POR (agent/por.py:73-112)
# ---------------------------------------------------------------------------------------------
@dataclass
class PorOracle:
    P: dict
    S: int
    H: int
    L: int
    layer_norm: bool = False
    tau: float = 0.9
    alpha: float = 10.0
    discount: float = 0.99
    beta: float = 0.005
    value_lr: float = 1e-4
    policy_lr: float = 1e-4
    max_steps: int = 1000
    vf: str = "vf"
    vt: str = "v_target"
    pol: str = "goal_policy"
    sorl: bool = False          # SORL: weight = exp(alpha*adv), tanh mean, NLL of actions
    sched_t: int = 0
    last_min_nlp: float = float("nan")

    def __post_init__(self):
        self.P = {k: np.array(v, dtype=F32, copy=True) for k, v in self.P.items()}
        self.vf_names = twin_param_names(self.vf, self.L, self.layer_norm)
        self.pol_names = [self.pol + ".log_std"] + mlp_param_names(self.pol + ".net", self.L, False)
        self.adam_v = AdamState(self.value_lr, self.vf_names)
        self.adam_g = AdamState(self.policy_lr, self.pol_names)

    def value_update(self, obs, next_obs, rew, term, inv_batch=None):
        P = self.P
        v_loss, target_v, G = value_step(P, self.vf, self.vt, self.adam_v, obs, next_obs, rew, term,
                                         self.L, self.layer_norm, self.tau, self.discount, self.beta,
                                         inv_batch)
        self.last_vf_grads = G
        adam_step(P, G, self.adam_v)                                          # por.py:88-90
        ema_update(P, self.vt, self.vf, self.vf_names, self.beta)             # por.py:93
        return v_loss, target_v

    def policy_update(self, obs, target_v, policy_target, inv_batch=None):
        P = self.P
        v1, v2, _, _ = twin_forward(P, self.vf, obs, self.L, self.layer_norm)  # updated vf, por.py:97
        adv = target_v - np.minimum(v1, v2)
        if self.sorl:
            weight = np.exp(F32(self.alpha) * adv)                            # sorl.py:104
        else:
            weight = np.exp(adv / F32(self.alpha))                            # por.py:100
        weight = np.minimum(weight, F32(EXP_ADV_MAX)).astype(F32)             # por.py:101
        out_act = "tanh" if self.sorl else None
        mean, cp = mlp_forward(P, self.pol + ".net", obs, self.L, False, out_act)
        nlp, z, sigma = gaussian_nll(mean, policy_target, P[self.pol + ".log_std"])
        self.last_min_nlp = float(nlp.min())                                  # por.py:104 trap value
        ib = F32(1.0 / obs.shape[0]) if inv_batch is None else F32(inv_batch)
        g_loss = float(np.sum(weight * nlp, dtype=F32) * ib)                  # por.py:106
        d_mean, d_ls = policy_nll_grads(z, sigma, P[self.pol + ".log_std"], weight * ib)
        G = mlp_backward(P, self.pol + ".net", cp, d_mean, self.L, False, out_act)
        G[self.pol + ".log_std"] = d_ls
        self.last_pol_grads = G
        lr = cosine_lr(self.policy_lr, self.sched_t, self.max_steps)          # Appendix A.3
        adam_step(P, G, self.adam_g, lr=lr)                                   # por.py:107-109
        self.sched_t += 1                                                     # por.py:110
        return g_loss

    def por_residual_update(self, obs, next_obs, rew, term):
        """agent/por.py:73-112 -> (v_loss, g_loss)."""
        obs, next_obs = np.ascontiguousarray(obs, F32), np.ascontiguousarray(next_obs, F32)
        rew, term = np.asarray(rew, F32), np.asarray(term, F32)
        v_loss, target_v = self.value_update(obs, next_obs, rew, term)
        g_loss = self.policy_update(obs, target_v, next_obs)
        return v_loss, g_loss

    def sorl_update(self, obs, actions, rew, next_obs, term):
        """agent/sorl.py:78-128 (backbone=None) -> (v_loss, g_loss)."""
        obs, next_obs = np.ascontiguousarray(obs, F32), np.ascontiguousarray(next_obs, F32)
        rew, term = np.asarray(rew, F32), np.asarray(term, F32)
        v_loss, target_v = self.value_update(obs, next_obs, rew, term)
        g_loss = self.policy_update(obs, target_v, np.ascontiguousarray(actions, F32))
        return v_loss, g_loss

    def sorl_vf_update(self, obs, actions, rew, next_obs, term):
        """agent/sorl.py:130-152 -> v_loss."""
        v_loss, _ = self.value_update(np.ascontiguousarray(obs, F32), np.ascontiguousarray(next_obs, F32),
                                      np.asarray(rew, F32), np.asarray(term, F32))
        return v_loss

    def select_action(self, obs):
        """agent/sorl.py:71-76: distribution mean."""
        mean, _ = mlp_forward(self.P, self.pol + ".net", np.ascontiguousarray(obs, F32), self.L, False,
                              "tanh" if self.sorl else None)
        return mean


def sorl_oracle(P, S, H, L, layer_norm=False, **kw):
    return PorOracle(P, S, H, L, layer_norm, vf="v_net", vt="v_tgt", pol="policy", sorl=True, **kw)


# ---------------------------------------------------------------------------------------------
# CQL (src/porl/train/cql_trainer.py:60-124; QNetwork src/porl/net/q_network.py:8-30)
# ---------------------------------------------------------------------------------------------
def qnet_names(prefix, n_layers):
    return [f"{prefix}model.{2 * i}.{p}" for i in range(n_layers) for p in ("weight", "bias")]


def qnet_forward(P, prefix, x, n_layers):
    acts = [x.astype(F32, copy=False)]
    h = acts[0]
    for i in range(n_layers):
        z = h @ P[f"{prefix}model.{2 * i}.weight"].T + P[f"{prefix}model.{2 * i}.bias"]
        h = np.maximum(z, F32(0.0)) if i < n_layers - 1 else z
        acts.append(h)
    return h, acts


def qnet_backward(P, prefix, acts, dq, n_layers):
    G = {}
    d = dq.astype(F32, copy=False)
    for i in reversed(range(n_layers)):
        if i < n_layers - 1:
            d = d * (acts[i + 1] > 0)
        G[f"{prefix}model.{2 * i}.weight"] = d.T @ acts[i]
        G[f"{prefix}model.{2 * i}.bias"] = d.sum(axis=0, dtype=F32)
        if i > 0:
            d = d @ P[f"{prefix}model.{2 * i}.weight"]
    return G


def logsumexp_rows(q):
    m = q.max(axis=1, keepdims=True)
    return (m[:, 0] + np.log(np.sum(np.exp(q - m), axis=1, dtype=F32))).astype(F32)


@dataclass
class CqlOracle:
    Q: dict                     # q_network params, keys 'model.0.weight' ...
    n_actions: int
    gamma: float = 0.99
    alpha: float = 1.0
    lr: float = 5e-4
    n_layers: int = 4           # 3 hidden (64,128,64) + output

    def __post_init__(self):
        self.Q = {k: np.array(v, dtype=F32, copy=True) for k, v in self.Q.items()}
        self.T = {k: v.copy() for k, v in self.Q.items()}     # target_network.load_state_dict
        self.adam = AdamState(self.lr, qnet_names("", self.n_layers))

    def penalty(self, states, actions):
        """compute_cql_penalty, cql_trainer.py:60-86."""
        q, _ = qnet_forward(self.Q, "", np.ascontiguousarray(states, F32), self.n_layers)
        lse = logsumexp_rows(q) - F32(math.log(self.n_actions))
        return float(np.mean(lse - q[np.arange(q.shape[0]), actions], dtype=F32))

    def learn(self, states, actions, rewards, next_states, dones):
        """cql_trainer.py:88-124 on an explicit minibatch -> loss."""
        B = states.shape[0]
        ar = np.arange(B)
        nq, _ = qnet_forward(self.T, "", np.ascontiguousarray(next_states, F32), self.n_layers)
        tq = nq.max(axis=1)                                                   # argmax+gather :95-97
        y = (rewards + F32(self.gamma) * tq * (F32(1.0) - dones)).astype(F32)  # :98
        q, acts = qnet_forward(self.Q, "", np.ascontiguousarray(states, F32), self.n_layers)
        qa = q[ar, actions]
        td = np.mean((qa - y) ** 2, dtype=F32)                                # F.mse_loss :102
        lse = logsumexp_rows(q)
        pen = np.mean(lse - F32(math.log(self.n_actions)) - qa, dtype=F32)    # :77-86
        loss = td + F32(self.alpha) * pen                                     # :108
        sm = np.exp(q - lse[:, None]).astype(F32)
        dq = (F32(self.alpha) / F32(B)) * sm
        dq[ar, actions] += F32(2.0 / B) * (qa - y) - F32(self.alpha) / F32(B)
        G = qnet_backward(self.Q, "", acts, dq, self.n_layers)
        adam_step(self.Q, G, self.adam)                                       # :111-113
        return float(loss)

    def sync_target(self):
        """dqn_trainer.py:195-196 hard copy."""
        self.T = {k: v.copy() for k, v in self.Q.items()}
