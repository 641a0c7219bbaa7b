"""CQL trainer — drop-in for the hot path of /root/reference/src/porl/train/cql_trainer.py (+ the pieces of
dqn_trainer.py it inherits): `learn()`, `compute_cql_penalty`, `train_offline`, `get_action`,
`select_action`, with `q_network`, `target_network`, `optimizer`, `replay_buffer`, `batch_size`, `gamma`,
`alpha`, `action_size`, `device` attributes.

The reference's constructor is broken upstream (it forwards 10 positionals to an 11-positional base,
SURVEY.md §2.1); this one accepts the keyword call of scripts/train_cql.py:18-29 and behaves as evidently
intended (`learning_rate` defaults to the hard-coded 5e-4 of dqn_trainer.py:71).

`learn()` = sample (host index stream identical to the reference under the same numpy seed, gather on the
device) + one fused device update: target forward, online forward, TD + CQL(H) penalty, backward, Adam —
hand-written gfx950 kernels behind `porl_qnet_*` (include/porl_hip.h).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn

from .. import _native as N
from ..buffer.replay_buffer import ReplayBuffer
from ..engine import _norm_device
from ..net.q_network import DuelingQNetwork, QNetwork
from ..parallel import GradExchange
from ..utils.logger import Logger


class QnetEngine:
    def __init__(self, state_dim, n_actions, hidden, max_batch, device):
        self.device = _norm_device(device)
        hid = (C.c_int32 * 8)(*hidden)
        self.cfg = N.QnetCfg(int(state_dim), int(n_actions), len(hidden), hid, int(max_batch))
        self._lib = N.lib()
        h = C.c_void_p()
        N.check(self._lib.porl_qnet_create(C.byref(self.cfg), C.byref(h)), "porl_qnet_create")
        self._h = h
        self.n_params = int(self._lib.porl_qnet_param_floats(h))
        z = lambda: torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.params, self.params_tgt, self.grads, self.adam_m, self.adam_v = z(), z(), z(), z(), z()
        self.stats = torch.zeros(8, dtype=torch.float32, device=self.device)
        self.workspace, self._bound = None, False

    @property
    def fused(self):
        """True when the network fits the one-launch step kernel (csrc/qnet_fused.hpp)."""
        return bool(self._lib.porl_qnet_one_launch(self._h))

    def tensor_table(self):
        """[(offset, shape, row stride)] — weight matrices sit inside zero-padded images (include/porl_hip.h)."""
        out = []
        off, r, c, ld = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
        for i in range(int(self._lib.porl_qnet_tensors(self._h))):
            N.check(self._lib.porl_qnet_tensor_info(self._h, i, C.byref(off), C.byref(r), C.byref(c), C.byref(ld)))
            out.append((off.value, (c.value,) if r.value == 0 else (r.value, c.value), ld.value))
        return out

    def views(self, flat):
        """Parameter-shaped views into a flat group (weights: strided (out, in) windows of their padded images)."""
        return [flat[o:o + s[0]] if len(s) == 1 else flat[o:o + s[0] * ld].view(s[0], ld)[:, :s[1]]
                for o, s, ld in self.tensor_table()]

    def _ensure_bound(self):
        if self.device.type != "cuda":
            raise N.NativeError("porl_amd computes on a HIP device only (device='cuda'); there is no CPU path")
        if self._bound:
            return
        self.workspace = torch.empty(int(self._lib.porl_qnet_workspace_floats(self._h)), dtype=torch.float32,
                                     device=self.device)
        b = N.QnetBuffers(*[C.c_void_p(t.data_ptr()) for t in (self.params, self.params_tgt, self.grads,
                                                                 self.adam_m, self.adam_v, self.workspace, self.stats)])
        N.check(self._lib.porl_qnet_bind(self._h, C.byref(b)), "porl_qnet_bind")
        self._bound = True

    def _states(self, x, what="states"):
        if x.dtype != torch.float32:
            x = x.float()
        x = x.reshape(x.shape[0], -1)
        if x.shape[1] != self.cfg.state_dim:
            raise RuntimeError(f"{what}: expected (B, {self.cfg.state_dim}), got {tuple(x.shape)}")
        if x.device != self.device:
            raise RuntimeError(f"{what} on {x.device}, engine on {self.device}")
        return x if x.stride(1) == 1 else x.contiguous()

    def load_batch(self, states, actions, rewards, next_states, dones):
        self._ensure_bound()
        states, next_states = self._states(states), self._states(next_states, "next_states")
        B = states.shape[0]
        if B > self.cfg.max_batch:
            raise RuntimeError(f"batch {B} exceeds engine max_batch {self.cfg.max_batch}")
        if actions.dtype != torch.int64:
            actions = actions.long()
        rewards, dones = rewards.float(), dones.float()
        self._held = (states, actions, rewards, next_states, dones)
        N.check(self._lib.porl_qnet_load_batch(self._h, B, N.ptr(states), states.stride(0), N.ptr(actions),
                                               actions.stride(0), N.ptr(rewards), rewards.stride(0),
                                               N.ptr(next_states), next_states.stride(0), N.ptr(dones),
                                               dones.stride(0), N.current_stream_ptr(self.device)), "porl_qnet_load_batch")
        return B

    def hyper(self, gamma, alpha, inv_batch, step, lr, betas=(0.9, 0.999), eps=1e-8):
        return N.QnetHyper(gamma, alpha, inv_batch, step, lr, betas[0], betas[1], eps)

    post_step = None      # callable(hp) run after every optimizer step of this engine (dueling heads: _DuelingHeads)

    def _stepped(self, hp):
        if self.post_step is not None:
            self.post_step(hp)

    def cql_backward(self, hp): N.check(self._lib.porl_qnet_cql_backward(self._h, C.byref(hp), N.current_stream_ptr(self.device)), "porl_qnet_cql_backward")

    def apply(self, hp):
        N.check(self._lib.porl_qnet_apply(self._h, C.byref(hp), N.current_stream_ptr(self.device)), "porl_qnet_apply")
        self._stepped(hp)

    def learn(self, hp):
        N.check(self._lib.porl_qnet_learn(self._h, C.byref(hp), N.current_stream_ptr(self.device)), "porl_qnet_learn")
        self._stepped(hp)

    def learn_indexed(self, hp, states, actions, rewards, next_states, dones, idx, variant=None):
        """learn() on rows `idx` (int64, device) of device-resident replay arrays: gathered inside the one-launch step
        kernel, or — networks with a layer wider than 128 or more than five Linear layers — by one gather launch in
        front of the multi-launch path (same loss arithmetic, incl. the Double-DQN / importance-weight / BCQ-mask
        variants)."""
        self._ensure_bound()
        B = idx.numel()
        if B > self.cfg.max_batch:
            raise RuntimeError(f"batch {B} exceeds engine max_batch {self.cfg.max_batch}")
        for name, x, dt in (("states", states, torch.float32), ("next_states", next_states, torch.float32),
                            ("actions", actions, torch.int64), ("rewards", rewards, torch.float32),
                            ("dones", dones, torch.float32), ("idx", idx, torch.int64)):
            if x.dtype != dt or x.device != self.device or not x.is_contiguous():
                raise RuntimeError(f"{name}: need a contiguous {dt} tensor on {self.device}")
        if states.shape[1:].numel() != self.cfg.state_dim or next_states.shape != states.shape:
            raise RuntimeError("replay arrays do not match the network's state_dim")
        if variant is None:
            N.check(self._lib.porl_qnet_learn_indexed(self._h, N.ptr(states), self.cfg.state_dim, N.ptr(actions),
                                                      N.ptr(rewards), N.ptr(next_states), self.cfg.state_dim, N.ptr(dones),
                                                      N.ptr(idx), B, C.byref(hp), N.current_stream_ptr(self.device)),
                    "porl_qnet_learn_indexed")
        else:
            N.check(self._lib.porl_qnet_learn_variant(self._h, N.ptr(states), self.cfg.state_dim, N.ptr(actions),
                                                      N.ptr(rewards), N.ptr(next_states), self.cfg.state_dim, N.ptr(dones),
                                                      N.ptr(idx), B, C.byref(hp), C.byref(variant),
                                                      N.current_stream_ptr(self.device)), "porl_qnet_learn_variant")
        self._stepped(hp)
        return B

    def learn_sampled(self, hp, states, actions, rewards, next_states, dones, n_rows, batch, seed, draw):
        """learn() on `batch` distinct rows of the first `n_rows` rows of device-resident replay arrays, drawn inside
        the step kernel (the indices of engine.sample_indices(n_rows, batch, seed, draw), never materialised)."""
        self._ensure_bound()
        if batch > self.cfg.max_batch:
            raise RuntimeError(f"batch {batch} exceeds engine max_batch {self.cfg.max_batch}")
        for name, x, dt in (("states", states, torch.float32), ("next_states", next_states, torch.float32),
                            ("actions", actions, torch.int64), ("rewards", rewards, torch.float32),
                            ("dones", dones, torch.float32)):
            if x.dtype != dt or x.device != self.device or not x.is_contiguous() or x.shape[0] < n_rows:
                raise RuntimeError(f"{name}: need a contiguous {dt} tensor of >= {n_rows} rows on {self.device}")
        if states.shape[1:].numel() != self.cfg.state_dim or next_states.shape != states.shape:
            raise RuntimeError("replay arrays do not match the network's state_dim")
        N.check(self._lib.porl_qnet_learn_sampled(self._h, N.ptr(states), self.cfg.state_dim, N.ptr(actions), N.ptr(rewards),
                                                  N.ptr(next_states), self.cfg.state_dim, N.ptr(dones), int(n_rows),
                                                  int(seed), int(draw), int(batch), C.byref(hp),
                                                  N.current_stream_ptr(self.device)), "porl_qnet_learn_sampled")
        self._stepped(hp)
        return batch

    @property
    def can_sample(self):
        return bool(self._lib.porl_qnet_can_sample(self._h))

    post_sync = None      # callable() run after the target network was overwritten (dueling heads)

    def sync_target(self):
        self._ensure_bound()
        N.check(self._lib.porl_qnet_sync_target(self._h, N.current_stream_ptr(self.device)), "porl_qnet_sync_target")
        if self.post_sync is not None:
            self.post_sync()

    def forward(self, x, which=0):
        self._ensure_bound()
        x = self._states(x)
        q = torch.empty(x.shape[0], self.cfg.n_actions, dtype=torch.float32, device=self.device)
        N.check(self._lib.porl_qnet_forward(self._h, which, N.ptr(x), x.stride(0), x.shape[0], N.ptr(q),
                                            self.cfg.n_actions, N.current_stream_ptr(self.device)), "porl_qnet_forward")
        return q

    def penalty(self, states, actions):
        self._ensure_bound()
        states = self._states(states)
        actions = actions.long()
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        N.check(self._lib.porl_qnet_penalty(self._h, N.ptr(states), states.stride(0), N.ptr(actions),
                                            actions.stride(0), states.shape[0], N.ptr(out),
                                            N.current_stream_ptr(self.device)), "porl_qnet_penalty")
        return out[0]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.porl_qnet_destroy(self._h)
                self._h = None
        except Exception:
            pass


class _DuelingHeads:
    """True parameters of a DuelingQNetwork pair's heads (reference q_network.py:41-49) beside an engine whose output layer
    is the COMPOSED layer  W_eff = M [w_v; W_a],  b_eff = M [b_v; b_a]  with the constant M = [1 | I - 11^T/A] (A, A+1).

    Storage per network: one flat (A+1, F+1) fp32 tensor, row 0 = [w_v | b_v], rows 1.. = [W_a[j] | b_a[j]]; the module's
    `value.0.*` / `advantage.0.*` parameters are (strided) views of it, Adam moments mirror it.  After every optimizer step
    of the engine: gather the output layer's gradient (A, F+1) from the engine's gradient group, map it back with
    M^T (one TN product on porl_gemm_f32), torch-exact Adam on the heads (porl_adam_ema, the trainer's step counter and
    hyper-parameters), compose (one NN product) and write the composed layer into the engine's parameter image.  The
    engine's own Adam also steps the composed layer — overwritten right away, its moments are never used.  All arithmetic
    is HIP kernels; torch only copies between strided views."""

    def __init__(self, trainer, q_network, target_network):
        eng = trainer._engine
        self.trainer, self.eng = trainer, eng
        A, F = eng.cfg.n_actions, 64
        self.A, self.F = A, F
        dev = eng.device
        n = ((A + 1) * (F + 1) + 3) // 4 * 4
        z = lambda: torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat, self.flat_tgt, self.grad, self.m, self.v = z(), z(), z(), z(), z()
        M = torch.zeros(A, A + 1, dtype=torch.float32)
        M[:, 0] = 1.0
        M[:, 1:] = torch.eye(A) - 1.0 / A
        self.M = M.to(dev)
        self.g_ext = torch.zeros(A, F + 1, dtype=torch.float32, device=dev)
        self.w_ext = torch.zeros(A, F + 1, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for mod, flat in ((q_network, self.flat), (target_network, self.flat_tgt)):
                hd = flat[:(A + 1) * (F + 1)].view(A + 1, F + 1)
                for p, v in ((mod.value[0].weight, hd[0:1, :F]), (mod.value[0].bias, hd[0, F:F + 1]),
                             (mod.advantage[0].weight, hd[1:, :F]), (mod.advantage[0].bias, hd[1:, F])):
                    v.copy_(p)
                    p.data = v
        self.compose(0)
        self.flat_tgt_sync = False

    def _hd(self, flat):
        return flat[:(self.A + 1) * (self.F + 1)].view(self.A + 1, self.F + 1)

    def compose(self, which):
        """Write the composed output layer of the online (0) / target (1) network into the engine's parameter image."""
        from .. import engine as E
        A, F, eng = self.A, self.F, self.eng
        hd = self._hd(self.flat if which == 0 else self.flat_tgt)
        E.gemm_f32(1, self.M, hd, A, F + 1, A + 1, A + 1, F + 1, self.w_ext, F + 1)          # (A, A+1) x (A+1, F+1)
        views = eng.views(eng.params if which == 0 else eng.params_tgt)
        with torch.no_grad():
            views[-2].copy_(self.w_ext[:, :F])
            views[-1].copy_(self.w_ext[:, F])

    def after_step(self, hp):
        from .. import engine as E
        A, F, eng = self.A, self.F, self.eng
        gv = eng.views(eng.grads)
        with torch.no_grad():
            self.g_ext[:, :F].copy_(gv[-2])
            self.g_ext[:, F].copy_(gv[-1])
        # d(heads) = M^T d(composed layer): "TN" product, contraction over the A composed rows
        E.gemm_f32(2, self.M, self.g_ext, A + 1, F + 1, A, A + 1, F + 1, self._hd(self.grad), F + 1)
        g = self.trainer.optimizer.param_groups[0]
        E.adam_ema(self.flat, self.grad, self.m, self.v, None, hp.lr, hp.step, g["betas"][0], g["betas"][1], g["eps"], 0.0)
        self.compose(0)

    def after_sync(self):
        with torch.no_grad():
            self.flat_tgt.copy_(self.flat)
        # (the engine copied the online image, composed layer included, into the target image)


class _FlatAdam:
    """torch.optim.Adam-format state over the engine's flat group (see agent/_iql.py:ArenaAdam)."""

    def __init__(self, engine, params, lr):
        self._eng, self._params, self.step_count = engine, params, 0
        self.param_groups = [dict(lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, params=params)]

    def zero_grad(self, set_to_none=True):
        pass

    _dueling = None       # _DuelingHeads of a DuelingQNetwork trainer: the heads' moments live beside the engine's

    def _moments(self):
        ms, vs = self._eng.views(self._eng.adam_m), self._eng.views(self._eng.adam_v)
        if self._dueling is not None:
            # q_network.parameters() order: value.w, value.b, advantage.w, advantage.b, then the hidden layers (the engine's
            # last two tensors are the composed output layer, which is not a parameter)
            d = self._dueling
            hm, hv = d._hd(d.m), d._hd(d.v)
            pick = lambda h: [h[0:1, :d.F], h[0, d.F:d.F + 1], h[1:, :d.F], h[1:, d.F]]
            return pick(hm) + ms[:-2], pick(hv) + vs[:-2]
        return ms, vs

    def state_dict(self):
        ms, vs = self._moments()
        state = {i: dict(step=torch.tensor(float(self.step_count)), exp_avg=m.clone(), exp_avg_sq=v.clone())
                 for i, (m, v) in enumerate(zip(ms, vs))} if self.step_count else {}
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g["params"] = list(range(len(self._params)))
        return dict(state=state, param_groups=[g])


class CQLTrainer:
    def __init__(self, state_size, action_size, gamma, epsilon=1.0, epsilon_min=0.05, epsilon_decay=0.99,
                 update_target_freq=10, device=torch.device("cpu"), network=QNetwork, log_dir="logs",
                 num_epochs=1000, threshold=0.1, alpha=1, learning_rate=0.0005, replay_buffer=None,
                 batch_size=64, max_batch=4096):
        self.state_size, self.action_size = state_size, action_size
        self.device = torch.device(device)
        self.gamma, self.epsilon, self.epsilon_min, self.epsilon_decay = gamma, epsilon, epsilon_min, epsilon_decay
        self.learning_rate, self.update_target_freq = learning_rate, update_target_freq
        self.num_epochs, self.threshold, self.alpha = num_epochs, threshold, alpha
        # same construction order / RNG consumption as dqn_trainer.py:66-70; `network` is any callable (state_size,
        # action_size) -> QNetwork, e.g. `lambda s, a: QNetwork(s, a, [256, 256])` for other hidden sizes
        self.q_network = network(state_size, action_size)
        self.target_network = network(state_size, action_size)
        dueling = isinstance(self.q_network, DuelingQNetwork) and isinstance(self.target_network, DuelingQNetwork)
        if not dueling and (not isinstance(self.q_network, QNetwork) or not isinstance(self.target_network, QNetwork)):
            raise NotImplementedError("QNetwork (a plain Linear/ReLU chain) and DuelingQNetwork are on the accelerated path")
        hidden = self.q_network._spec[2]
        self._engine = QnetEngine(state_size, action_size, hidden, max(max_batch, batch_size), self.device)
        with torch.no_grad():
            for mod, flat, which in ((self.q_network, self._engine.params, 0),
                                     (self.target_network, self._engine.params_tgt, 1)):
                # dueling: `model` holds the hidden layers only; the engine's output layer is the composed layer
                owned = list(mod.model.parameters()) if dueling else list(mod.parameters())
                for p, v in zip(owned, self._engine.views(flat)):
                    v.copy_(p)
                    p.data = v
                mod._engine, mod._which = self._engine, which
            if not dueling:
                self._engine.params_tgt.copy_(self._engine.params)   # target.load_state_dict(q.state_dict())
        self._dueling = None
        if dueling:
            self._engine._ensure_bound()
            self._dueling = _DuelingHeads(self, self.q_network, self.target_network)
            # target.load_state_dict(q.state_dict()) (dqn_trainer.py:68): hidden layers, composed layer and heads
            self._engine.params_tgt.copy_(self._engine.params)
            self._dueling.after_sync()
            self._engine.post_step, self._engine.post_sync = self._dueling.after_step, self._dueling.after_sync
            for mod, which in ((self.q_network, 0), (self.target_network, 1)):
                mod.register_load_state_dict_post_hook(lambda module, incompatible, w=which: self._dueling.compose(w))
        self.target_network.eval()
        self.optimizer = _FlatAdam(self._engine, list(self.q_network.parameters()), learning_rate)
        self.optimizer._dueling = self._dueling
        self.replay_buffer = replay_buffer if replay_buffer is not None else ReplayBuffer(100000, (state_size,), self.device)
        self.batch_size = batch_size
        self.logger = Logger(log_dir=log_dir)
        self.logger.log_hyperparameters(dict(learning_rate=learning_rate, gamma=gamma, batch_size=batch_size,
                                             epsilon_decay=epsilon_decay, update_target_freq=update_target_freq))
        self.training_step = 0
        self._exchange = GradExchange()
        self.async_losses = False

    # -- hot path ---------------------------------------------------------------------------------
    def learn_on(self, states, actions, rewards, next_states, dones):
        """cql_trainer.py:94-124 on an explicit minibatch (device tensors)."""
        eng, ex = self._engine, self._exchange
        B = eng.load_batch(states, actions, rewards, next_states, dones)
        self.optimizer.step_count += 1
        g = self.optimizer.param_groups[0]
        hp = eng.hyper(self.gamma, float(self.alpha), 1.0 / (B * ex.world_size), self.optimizer.step_count,
                       g["lr"], g["betas"], g["eps"])
        if not ex.active:
            eng.learn(hp)
        else:
            eng.cql_backward(hp)
            ex.allreduce_sum_(eng.grads)
            ex.allreduce_sum_(eng.stats[:3])
            eng.apply(hp)
        if self.async_losses:
            return eng.stats[:3]
        loss, self.last_td_loss, self.last_cql_penalty = eng.stats[:3].tolist()
        return loss

    def learn(self):
        return self.learn_on(*self.replay_buffer.sample(self.batch_size))

    def learn_device_sampled(self, seed=0):
        """Extension (not in the reference): `learn()` with the B distinct indices drawn on the device (keyed
        permutation) instead of numpy's O(N) host permutation — no host work, no host->device copies."""
        from .. import engine as E
        rb = self.replay_buffer
        if self.batch_size > rb.size:
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
        rb._sync_mirror()
        self._draws = getattr(self, "_draws", 0)
        draw = self._draws
        self._draws += 1
        eng, ex = self._engine, self._exchange
        if ex.active or not eng.fused:
            idx = E.sample_indices(rb.size, self.batch_size, seed, draw, device=self.device)
            return self.learn_on(*rb.gather_device(idx))
        # one-launch path: the step kernel draws the rows itself (or gathers rows idx of the device mirror)
        m = rb._mirror
        self.optimizer.step_count += 1
        g = self.optimizer.param_groups[0]
        hp = eng.hyper(self.gamma, float(self.alpha), 1.0 / self.batch_size, self.optimizer.step_count, g["lr"], g["betas"],
                       g["eps"])
        if eng.can_sample:
            eng.learn_sampled(hp, m["states"], m["actions"], m["rewards"], m["next_states"], m["dones"], rb.size,
                              self.batch_size, seed, draw)
        else:
            idx = E.sample_indices(rb.size, self.batch_size, seed, draw, device=self.device)
            eng.learn_indexed(hp, m["states"], m["actions"], m["rewards"], m["next_states"], m["dones"], idx)
        if self.async_losses:
            return eng.stats[:3]
        loss, self.last_td_loss, self.last_cql_penalty = eng.stats[:3].tolist()
        return loss

    def compute_cql_penalty(self, states, actions):
        return self._engine.penalty(states, actions)

    def sync_target(self):
        self._engine.sync_target()

    def train_offline(self, policy=None, num_iterations: int = 10000):
        losses = []
        for self.training_step in range(num_iterations):
            loss = self.learn() if policy is None else policy()
            self.logger.log_loss(self.training_step, loss)
            if self.training_step % self.update_target_freq == 0:      # dqn_trainer.py:195-196
                self.sync_target()
            losses.append(loss)
        self.logger.close()
        return losses

    # -- acting -----------------------------------------------------------------------------------
    def get_action(self, state: np.ndarray) -> int:
        x = torch.as_tensor(np.asarray(state), dtype=torch.float32, device=self.device).unsqueeze(0)
        return int(self.q_network(x).argmax(dim=1).item())

    def select_action(self, state: np.ndarray) -> int:
        if np.random.rand() < self.epsilon:
            return int(np.random.randint(self.action_size))
        return self.get_action(state)
