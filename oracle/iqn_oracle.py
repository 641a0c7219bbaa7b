"""TEST INFRASTRUCTURE — CPU restatement (numpy, fp64 by default) of the reference's Implicit Quantile Network and of one
IQNTrainer.learn step.  Only tests/ may import this module; the product path (porl_amd/) never does.

Follows /root/reference/src/porl/net/iqn_network.py:35-91 (forward, get_quantile_embedding) and
/root/reference/src/porl/train/iqn_trainer.py:92-149 (learn, quantile_huber_loss) with `get_q_values` read as the live
network's forward (porl_amd/train/iqn_trainer.py explains why).  Pinned by tests/golden/iqn_s9_a5.npz, which
oracle/gen_golden.py:gen_iqn produces by running upstream's own learn() on the reference network.

Parameters: a dict of arrays under the reference's state_dict names
(feature_net.{0,2}.{weight,bias}, quantile_embedding.{weight,bias}, value_net.{0,2}.{weight,bias}).
"""
import numpy as np

NAMES = ["feature_net.0.weight", "feature_net.0.bias", "feature_net.2.weight", "feature_net.2.bias",
         "quantile_embedding.weight", "quantile_embedding.bias",
         "value_net.0.weight", "value_net.0.bias", "value_net.2.weight", "value_net.2.bias"]


def cos_embed(taus, E):
    """iqn_network.py:74-91: cos(pi * i * tau), i = 1..E -> (B, N, E)"""
    i = np.arange(1, E + 1, dtype=taus.dtype).reshape(1, 1, -1)
    return np.cos(np.pi * i * taus[..., None])


def forward(P, states, taus, keep=False):
    """iqn_network.py:35-72 -> (B, N, A); keep=True also returns what backward() needs"""
    E = P["quantile_embedding.weight"].shape[1]
    h0 = np.maximum(states @ P["feature_net.0.weight"].T + P["feature_net.0.bias"], 0.0)
    feat = np.maximum(h0 @ P["feature_net.2.weight"].T + P["feature_net.2.bias"], 0.0)               # (B, H)
    ce = cos_embed(taus, E)                                                                          # (B, N, E)
    emb = ce @ P["quantile_embedding.weight"].T + P["quantile_embedding.bias"]                       # (B, N, H)
    comb = feat[:, None, :] * emb
    v0 = np.maximum(comb @ P["value_net.0.weight"].T + P["value_net.0.bias"], 0.0)
    z = v0 @ P["value_net.2.weight"].T + P["value_net.2.bias"]                                       # (B, N, A)
    return (z, (states, h0, feat, ce, emb, comb, v0)) if keep else z


def backward(P, cache, dz):
    """gradients of sum(z * dz) with respect to every parameter"""
    states, h0, feat, ce, emb, comb, v0 = cache
    G = {}
    G["value_net.2.weight"] = np.einsum("bna,bnh->ah", dz, v0)
    G["value_net.2.bias"] = dz.sum((0, 1))
    dv0 = (dz @ P["value_net.2.weight"]) * (v0 > 0)
    G["value_net.0.weight"] = np.einsum("bnk,bnh->kh", dv0, comb)
    G["value_net.0.bias"] = dv0.sum((0, 1))
    dcomb = dv0 @ P["value_net.0.weight"]
    demb = dcomb * feat[:, None, :]
    dfeat = (dcomb * emb).sum(1)
    G["quantile_embedding.weight"] = np.einsum("bnh,bne->he", demb, ce)
    G["quantile_embedding.bias"] = demb.sum((0, 1))
    d1 = dfeat * (feat > 0)
    G["feature_net.2.weight"] = d1.T @ h0
    G["feature_net.2.bias"] = d1.sum(0)
    d0 = (d1 @ P["feature_net.2.weight"]) * (h0 > 0)
    G["feature_net.0.weight"] = d0.T @ states
    G["feature_net.0.bias"] = d0.sum(0)
    return G


def quantile_huber(cur, td, taus, kappa):
    """iqn_trainer.py:127,136-149: loss and dloss/dcur; cur (B, N'), td (B, N''), taus (B, N')"""
    B, n_cur = cur.shape
    n_tgt = td.shape[1]
    u = td[:, None, :] - cur[:, :, None]                                                             # (B, N', N'')
    au = np.abs(u)
    hub = np.where(au <= kappa, 0.5 * u * u, kappa * (au - 0.5 * kappa))
    w = np.abs(taus[:, :, None] - (u < 0))
    loss = (w * hub).mean(2).mean(1).mean()
    dhub = np.where(au <= kappa, u, kappa * np.sign(u))
    dcur = -(w * dhub).sum(2) / (n_tgt * n_cur * B)
    return loss, dcur


class IqnOracle:
    def __init__(self, P, P_target, gamma, kappa, lr=5e-4, max_norm=10.0, dtype=np.float64):
        self.P = {k: np.asarray(v, dtype=dtype).copy() for k, v in P.items()}
        self.T = {k: np.asarray(v, dtype=dtype).copy() for k, v in P_target.items()}
        self.gamma, self.kappa, self.lr, self.max_norm, self.dtype = gamma, kappa, lr, max_norm, dtype
        self.m = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.P.items()}
        self.t = 0

    def learn(self, states, actions, rewards, next_states, dones, taus_p, taus_pp):
        """iqn_trainer.py:92-134 -> loss"""
        f = lambda a: np.asarray(a, dtype=self.dtype)
        states, rewards, next_states, dones, taus_p, taus_pp = map(f, (states, rewards, next_states, dones, taus_p, taus_pp))
        actions = np.asarray(actions).astype(np.int64).reshape(-1)
        B = states.shape[0]
        z, cache = forward(self.P, states, taus_p, keep=True)                                        # :98
        cur = z[np.arange(B), :, actions]                                                            # :101-103 (B, N')
        zo = forward(self.P, next_states, taus_pp)                                                   # :109
        a_star = zo.mean(1).argmax(1)                                                                # :110-111
        zt = forward(self.T, next_states, taus_pp)[np.arange(B), :, a_star]                          # :114-119
        td = rewards[:, None] + self.gamma * zt * (1.0 - dones[:, None])                             # :121
        loss, dcur = quantile_huber(cur, td, taus_p, self.kappa)
        dz = np.zeros_like(z)
        dz[np.arange(B), :, actions] = dcur
        G = backward(self.P, cache, dz)
        total = np.sqrt(sum(float((g * g).sum()) for g in G.values()))                               # :131 clip_grad_norm_
        coef = min(1.0, self.max_norm / (total + 1e-6))
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        for k in self.P:                                                                             # torch.optim.Adam
            g = G[k] * coef
            self.m[k] = b1 * self.m[k] + (1 - b1) * g
            self.v[k] = b2 * self.v[k] + (1 - b2) * g * g
            mh = self.m[k] / (1 - b1 ** self.t)
            vh = self.v[k] / (1 - b2 ** self.t)
            self.P[k] = self.P[k] - self.lr * mh / (np.sqrt(vh) + eps)
        self.grad_norm = total
        self.G = {k: G[k] * coef for k in G}             # what the optimizer saw (after clipping)
        return float(loss)
