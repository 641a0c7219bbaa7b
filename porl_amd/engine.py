"""Host-side owner of one IQL-family update engine (POR / SORL) on one MI355X.

PyTorch supplies device memory and the stream; all arithmetic of the step runs in libporl_hip.so
(porl_amd/_native.py).  Parameters, gradients and Adam moments each live in ONE flat fp32 tensor per
optimizer group so the Adam(+EMA) sweep and the data-parallel all-reduce are single operations; the
nn.Parameters of the agent modules are views into those tensors.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _native as N


def _norm_device(device):
    """torch.device('cuda') and torch.device('cuda:0') must compare equal for the checks below."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


class IqlEngine:
    GROUP_VF, GROUP_POL = 0, 1
    MODE_TWO_SLOTS, MODE_FOLD_COMBINE, MODE_SHORT_BLOCKS = 1, 2, 4     # include/porl_hip.h: PORL_IQL_MODE_*
    SLOTS = 3                                         # PORL_IQL_SLOTS: copies of the minibatch staging buffers

    def __init__(self, obs_dim, pol_out_dim, hidden_dim, n_hidden, layer_norm=False, pol_tanh=False,
                 weight_mode=0, max_batch=1024, device="cpu"):
        self.device = _norm_device(device)
        self.cfg = N.IqlCfg(int(obs_dim), int(pol_out_dim), int(hidden_dim), int(n_hidden),
                            int(bool(layer_norm)), int(bool(pol_tanh)), int(weight_mode), int(max_batch))
        self._lib = N.lib()
        h = C.c_void_p()
        N.check(self._lib.porl_iql_create(C.byref(self.cfg), C.byref(h)), "porl_iql_create")
        self._h = h
        self.n_vf = int(self._lib.porl_iql_group_floats(h, 0))
        self.n_pol = int(self._lib.porl_iql_group_floats(h, 1))
        self._bound = False
        self._mode = 0
        # pipelined updates (agent/_iql.py): the policy phase of update t runs on `side_stream` while the value phase
        # of update t+1 runs on the caller's stream; `_policy_done` is the event the side stream recorded last
        self._side = None
        self._policy_done = None
        self._values_read = None
        self._slot_users = [None] * self.SLOTS         # policy-done events of the last SLOTS pipelined updates, oldest first
        self._events = []
        self._alloc()

    # -- streams -----------------------------------------------------------------------------------
    def side_stream(self):
        if self._side is None:
            import os
            self._side = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("PORL_SIDE_PRIORITY", "0")))
        return self._side

    def event(self, i):
        """Small ring of reusable events (re-recording an event does not disturb waits already enqueued on it)."""
        while len(self._events) <= i:
            self._events.append(torch.cuda.Event(enable_timing=False, blocking=False))
        return self._events[i]

    # Cross-stream ordering of the pipelined update by counters in signal memory (csrc: porl_signal_*).  Three counters:
    # 0 = value Adam done (update number), 1 = policy forward half done, 2 = policy phase done (written and waited for
    # only under PORL_PIPE_SYNC=signal; the default, signal2, relies on the in-order side stream for slot reuse).
    # PORL_PIPE_SYNC=event keeps the event record / stream-wait-event pairs (A/B).
    SIG_VALUE, SIG_FWD, SIG_POLICY = 0, 1, 2

    def signals(self):
        """The three counters, or None when the device has no stream wait-value operations (events are used then)."""
        if getattr(self, "_signals", None) is None:
            sig = []
            with torch.cuda.device(self.device):
                for _ in range(3):
                    p = C.c_void_p()
                    if self._lib.porl_signal_create(C.byref(p)) != 0:
                        sig = False
                        break
                    sig.append(p)
            self._signals = sig
            self._seq = 0
        return self._signals or None

    def signal(self, which, value, stream):
        N.check(self._lib.porl_signal_write(self.signals()[which], int(value), C.c_void_p(stream.cuda_stream)), "porl_signal_write")

    def wait_signal(self, which, value, stream):
        if value > 0:
            N.check(self._lib.porl_signal_wait_ge(self.signals()[which], int(value), C.c_void_p(stream.cuda_stream)),
                    "porl_signal_wait_ge")

    def join(self):
        """Order the current stream behind an outstanding policy phase on the side stream (no host wait)."""
        ev, self._policy_done, self._values_read = self._policy_done, None, None
        self._slot_users = [None] * self.SLOTS
        if ev is True:                                 # signal mode: no per-update event; order behind the side stream now
            ev = self.event(3 * self.SLOTS)
            ev.record(self.side_stream())
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def wait_slot_free(self):
        """The staging slots rotate: the slot the next load writes was last read by the policy phase SLOTS pipelined
        updates ago, i.e. the oldest of the SLOTS events kept here (practically always complete already)."""
        ev = self._slot_users[0]
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def wait_values_read(self):
        """Order the current stream behind the forward half of the outstanding policy phase: after it the value
        parameters may be overwritten (the rest of that phase only touches policy state and its own scratch)."""
        ev, self._values_read = self._values_read, None
        if isinstance(ev, tuple):                      # signal mode: ("sig", update number of that policy phase)
            self.wait_signal(self.SIG_FWD, ev[1], torch.cuda.current_stream(self.device))
        elif ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def set_mode(self, mode):
        if mode != self._mode:
            N.check(self._lib.porl_iql_set_mode(self._h, int(mode)), "porl_iql_set_mode")
            self._mode = mode

    # -- memory ------------------------------------------------------------------------------------
    # Flat groups are allocated a little longer than the C library needs (zero tail) so that they split into equal,
    # 16-byte-aligned slices for every data-parallel world size that divides 840 (1..8, 10, 12, ...): the
    # reduce-scatter / sharded Adam / all-gather exchange (agent/_iql.py) works on those slices.
    SLICE_QUANTUM = 4 * 840

    def _alloc(self):
        dev = self.device
        q = self.SLICE_QUANTUM
        z = lambda n: torch.zeros((n + q - 1) // q * q, dtype=torch.float32, device=dev)
        self.params_vf, self.params_tgt, self.params_pol = z(self.n_vf), z(self.n_vf), z(self.n_pol)
        self.grads_vf, self.grads_pol = z(self.n_vf), z(self.n_pol)
        self.adam_m_vf, self.adam_v_vf = z(self.n_vf), z(self.n_vf)
        self.adam_m_pol, self.adam_v_pol = z(self.n_pol), z(self.n_pol)
        self.stats = z(8)
        self.workspace = None
        self._bound = False

    def _ensure_bound(self):
        if self.device.type != "cuda":
            raise N.NativeError("porl_amd computes on a HIP device only (device='cuda'); there is no CPU path")
        if self._bound:
            return
        nws = int(self._lib.porl_iql_workspace_floats(self._h))
        self.workspace = torch.empty(nws, dtype=torch.float32, device=self.device)
        b = N.IqlBuffers(*[C.c_void_p(t.data_ptr()) for t in (
            self.params_vf, self.params_tgt, self.params_pol, self.grads_vf, self.grads_pol,
            self.adam_m_vf, self.adam_v_vf, self.adam_m_pol, self.adam_v_pol, self.workspace, self.stats)])
        N.check(self._lib.porl_iql_bind(self._h, C.byref(b)), "porl_iql_bind")
        self._bound = True
        self._mode = 0
        N.check(self._lib.porl_iql_set_mode(self._h, 0), "porl_iql_set_mode")

    def to(self, device):
        """Move every flat tensor; views handed out earlier must be re-created by the caller."""
        device = _norm_device(device)
        if device == self.device:
            return self
        if self.device.type == "cuda":                 # an engine still on the CPU has no streams, events or signals
            self.join()
            torch.cuda.synchronize(self.device)
            self._free_signals()
        self._side, self._events = None, []
        self._policy_done = self._values_read = None
        for name in ("params_vf", "params_tgt", "params_pol", "grads_vf", "grads_pol", "adam_m_vf",
                     "adam_v_vf", "adam_m_pol", "adam_v_pol", "stats"):
            setattr(self, name, getattr(self, name).to(device))
        self.device = device
        self.workspace = None
        self._bound = False
        return self

    def tensor_table(self, group):
        """[(offset, shape)] of a group's tensors in named_parameters() order."""
        n = int(self._lib.porl_iql_group_tensors(self._h, group))
        out = []
        off, r, c = C.c_int64(), C.c_int32(), C.c_int32()
        for i in range(n):
            N.check(self._lib.porl_iql_tensor_info(self._h, group, i, C.byref(off), C.byref(r), C.byref(c)))
            out.append((off.value, (c.value,) if r.value == 0 else (r.value, c.value)))
        return out

    @staticmethod
    def views(flat, table):
        return [flat[o:o + math.prod(shape)].view(shape) for o, shape in table]

    # -- step --------------------------------------------------------------------------------------
    @staticmethod
    def _mat(t, cols, what):
        if t.dtype != torch.float32:
            raise RuntimeError(f"{what}: expected float32, got {t.dtype}")
        if t.dim() != 2 or t.shape[1] != cols:
            raise RuntimeError(f"{what}: expected shape (B, {cols}), got {tuple(t.shape)}")
        if t.stride(1) != 1:
            t = t.contiguous()
        return t

    def _vec(self, t, B, what):
        if t.dim() != 1 or t.shape[0] != B:
            raise RuntimeError(f"{what}: expected shape ({B},), got {tuple(t.shape)}")
        return t if t.dtype == torch.float32 else t.float()

    def load_batch(self, obs, next_obs, rew, term, pol_target=None):
        self._ensure_bound()
        for t in (obs, next_obs, rew, term, pol_target):
            if t is not None and t.device != self.device:
                raise RuntimeError(f"batch tensor on {t.device}, engine on {self.device}")
        S, D = self.cfg.obs_dim, self.cfg.pol_out_dim
        obs, next_obs = self._mat(obs, S, "observations"), self._mat(next_obs, S, "next_observations")
        B = obs.shape[0]
        if next_obs.shape[0] != B:
            raise RuntimeError("observations / next_observations batch mismatch")
        if B > self.cfg.max_batch:
            raise RuntimeError(f"batch {B} exceeds engine max_batch {self.cfg.max_batch}")
        rew, term = self._vec(rew, B, "rewards"), self._vec(term, B, "terminals")
        if pol_target is not None:
            pol_target = self._mat(pol_target, D, "policy target")
        # keep references alive until the pack kernel has run (stream-ordered; torch caches the memory)
        self._held = (obs, next_obs, rew, term, pol_target)
        N.check(self._lib.porl_iql_load_batch(
            self._h, B, N.ptr(obs), obs.stride(0), N.ptr(next_obs), next_obs.stride(0),
            N.ptr(rew), rew.stride(0), N.ptr(term), term.stride(0),
            N.ptr(pol_target), 0 if pol_target is None else pol_target.stride(0),
            N.current_stream_ptr(self.device)), "porl_iql_load_batch")
        self.last_batch = B
        return B

    def load_batch_sampled(self, rows, batch, seed, step, act_dim, target_is_action, idx_out=None):
        """Draw `batch` distinct rows of a device-resident packed-row store and stage them (one kernel)."""
        self._ensure_bound()
        if rows.device != self.device or rows.dtype != torch.float32 or rows.dim() != 2 or rows.stride(1) != 1:
            raise RuntimeError("replay rows must be a 2-D fp32 tensor on the engine's device")
        N.check(self._lib.porl_iql_load_batch_sampled(
            self._h, batch, N.ptr(rows), rows.stride(0), rows.shape[0], act_dim, int(target_is_action),
            seed & 0xFFFFFFFFFFFFFFFF, step, N.ptr(idx_out), N.current_stream_ptr(self.device)), "porl_iql_load_batch_sampled")
        self.last_batch = batch
        return batch

    def set_stats(self, stats):
        """Point the loss statistics of the following updates at `stats` (>= 8 fp32 on the device)."""
        self._ensure_bound()
        if stats.device != self.device or stats.dtype != torch.float32 or stats.numel() < 8 or not stats.is_contiguous():
            raise RuntimeError("stats must be >= 8 contiguous fp32 on the engine's device")
        N.check(self._lib.porl_iql_set_stats(self._h, N.ptr(stats)), "porl_iql_set_stats")
        self.stats = stats

    def group(self, g):
        """(params, grads, exp_avg, exp_avg_sq, target or None) flat tensors of optimizer group g."""
        if g == self.GROUP_VF:
            return self.params_vf, self.grads_vf, self.adam_m_vf, self.adam_v_vf, self.params_tgt
        return self.params_pol, self.grads_pol, self.adam_m_pol, self.adam_v_pol, None

    def hyper(self, **kw):
        d = dict(tau=0.9, discount=0.99, alpha=10.0, ema_beta=0.005, inv_batch=1.0, value_lr=1e-4,
                 policy_lr=1e-4, value_step=1, policy_step=1, adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8)
        d.update(kw)
        return N.IqlHyper(**d)

    def _phase(self, name, hp):
        N.check(getattr(self._lib, name)(self._h, C.byref(hp), N.current_stream_ptr(self.device)), name)

    def value_backward(self, hp): self._phase("porl_iql_value_backward", hp)
    def value_apply(self, hp): self._phase("porl_iql_value_apply", hp)
    def policy_forward(self, hp): self._phase("porl_iql_policy_forward", hp)
    def policy_backward(self, hp): self._phase("porl_iql_policy_backward", hp)
    def policy_apply(self, hp): self._phase("porl_iql_policy_apply", hp)
    def step(self, hp): self._phase("porl_iql_step", hp)

    def update_pipelined(self, hp, replay, batch, seq, wait_policy_seq, wait_fwd_seq, write_policy):
        """One pipelined update on rows drawn from `replay` (PackedReplay): value phase on the current stream, policy
        phase on the side stream, ordered by the signal counters — include/porl_hip.h:porl_iql_update_pipelined."""
        rows = replay.rows
        if rows.device != self.device or rows.dtype != torch.float32 or rows.dim() != 2 or rows.stride(1) != 1:
            raise RuntimeError("replay rows must be a 2-D fp32 tensor on the engine's device")
        sig = self.signals()
        N.check(self._lib.porl_iql_update_pipelined(
            self._h, C.byref(hp), batch, N.ptr(rows), rows.stride(0), rows.shape[0], replay.act_dim,
            int(self.cfg.weight_mode == 1), replay.seed & 0xFFFFFFFFFFFFFFFF, replay.draws,
            sig[self.SIG_VALUE], sig[self.SIG_FWD], sig[self.SIG_POLICY], int(seq), int(wait_policy_seq),
            int(wait_fwd_seq), int(bool(write_policy)), N.current_stream_ptr(self.device),
            C.c_void_p(self.side_stream().cuda_stream)), "porl_iql_update_pipelined")
        self.last_batch = batch

    def policy_prefetch(self):
        N.check(self._lib.porl_iql_policy_prefetch(self._h, N.current_stream_ptr(self.device)), "porl_iql_policy_prefetch")

    # -- forward-only ------------------------------------------------------------------------------
    def forward_value(self, x, target=False):
        self._ensure_bound()
        self.join()
        x = self._mat(x, self.cfg.obs_dim, "state")
        B = x.shape[0]
        v1 = torch.empty(B, dtype=torch.float32, device=self.device)
        v2 = torch.empty_like(v1)
        N.check(self._lib.porl_iql_forward_value(self._h, int(target), N.ptr(x), x.stride(0), B, N.ptr(v1),
                                                 N.ptr(v2), N.current_stream_ptr(self.device)), "porl_iql_forward_value")
        return v1, v2

    def forward_policy(self, x):
        self._ensure_bound()
        self.join()
        x = self._mat(x, self.cfg.obs_dim, "obs")
        B, D = x.shape[0], self.cfg.pol_out_dim
        mean = torch.empty(B, D, dtype=torch.float32, device=self.device)
        N.check(self._lib.porl_iql_forward_policy(self._h, N.ptr(x), x.stride(0), B, N.ptr(mean), D,
                                                  N.current_stream_ptr(self.device)), "porl_iql_forward_policy")
        return mean

    SMALL_BATCH = 8                                       # kernels.hpp: SMALL_FWD_MAX_B

    def forward_policy_host(self, x):
        """Rollout path (reference test.py:28-30: one observation in, one action out as numpy): the three small-batch
        launches read the observation from and write the mean to PINNED HOST memory — no copy launches, no torch
        kernels, one stream synchronisation.  `x`: (B <= 8, obs_dim) float32, a device tensor, a CPU tensor or an
        ndarray.  Returns a fresh (B, D) float32 ndarray."""
        self._ensure_bound()
        self.join()
        B, D, S = int(x.shape[0]), self.cfg.pol_out_dim, self.cfg.obs_dim
        if B < 1 or B > self.SMALL_BATCH or tuple(x.shape) != (B, S):
            raise RuntimeError(f"obs: expected shape (1..{self.SMALL_BATCH}, {S}), got {tuple(x.shape)}")
        if getattr(self, "_pin", None) is None:
            self._pin = (torch.empty(self.SMALL_BATCH, S, dtype=torch.float32).pin_memory(),
                         torch.empty(self.SMALL_BATCH, D, dtype=torch.float32).pin_memory())
        pin_in, pin_out = self._pin
        if isinstance(x, torch.Tensor) and x.device.type == "cuda":
            if x.dtype != torch.float32:
                raise RuntimeError(f"obs: expected float32, got {x.dtype}")
            src = x if x.stride(1) == 1 else x.contiguous()
            xp, xrs = N.ptr(src), src.stride(0)
        else:
            pin_in[:B].copy_(torch.as_tensor(x, dtype=torch.float32))      # host memcpy into the staging rows
            xp, xrs = N.ptr(pin_in), S
        stream = torch.cuda.current_stream(self.device)
        N.check(self._lib.porl_iql_forward_policy(self._h, xp, xrs, B, N.ptr(pin_out), D, C.c_void_p(stream.cuda_stream)),
                "porl_iql_forward_policy")
        stream.synchronize()
        return pin_out[:B].numpy().copy()

    def _free_signals(self):
        for p in (getattr(self, "_signals", None) or []):
            self._lib.porl_signal_destroy(p)
        self._signals = None

    def __del__(self):
        try:
            self._free_signals()
            if getattr(self, "_h", None):
                self._lib.porl_iql_destroy(self._h)
                self._h = None
        except Exception:
            pass


# -- thin wrappers over the building blocks ---------------------------------------------------------
def gemm_f32(mode, A, B, M, N_, K, lda, ldb, C_out, ldc, bias=None, act=0, mask=None, ldmask=0, tile=-1,
             splitk=1, slab=None):
    """Test/utility entry: raw pointers of torch tensors, see include/porl_hip.h:porl_gemm_f32."""
    N.check(N.lib().porl_gemm_f32(mode, tile, M, N_, K, N.ptr(A), lda, N.ptr(B), ldb, N.ptr(C_out), ldc,
                                  N.ptr(bias), act, N.ptr(mask), ldmask, splitk, N.ptr(slab),
                                  N.current_stream_ptr(C_out)), "porl_gemm_f32")


def adam_ema(p, g, m, v, target, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, ema_beta=0.0):
    N.check(N.lib().porl_adam_ema(N.ptr(p), N.ptr(g), N.ptr(m), N.ptr(v), N.ptr(target), p.numel(), lr, step,
                                  beta1, beta2, eps, ema_beta, N.current_stream_ptr(p)), "porl_adam_ema")


def ema(target, source, ema_beta):
    N.check(N.lib().porl_ema(N.ptr(target), N.ptr(source), target.numel(), ema_beta, N.current_stream_ptr(target)), "porl_ema")


def gather_rows(rows, idx, out=None):
    """out[i] = rows[idx[i]] for a 2-D fp32 (or bit-reinterpreted) device tensor."""
    if rows.dim() != 2 or rows.stride(1) != 1:
        raise RuntimeError("rows must be 2-D with unit column stride")
    n, w = idx.numel(), rows.shape[1]
    if out is None:
        out = torch.empty(n, w, dtype=rows.dtype, device=rows.device)
    N.check(N.lib().porl_gather_rows(N.ptr(rows), rows.stride(0), N.ptr(idx), n, w, N.ptr(out), out.stride(0),
                                     N.current_stream_ptr(out)), "porl_gather_rows")
    return out


def sample_indices(n_rows, batch, seed, step, out=None, base=0, device="cuda"):
    """`batch` distinct row indices in [base, base + n_rows) drawn on the device (int64)."""
    if out is None:
        out = torch.empty(batch, dtype=torch.int64, device=device)
    N.check(N.lib().porl_sample_indices(n_rows, batch, seed, step, base, N.ptr(out), N.current_stream_ptr(out)),
            "porl_sample_indices")
    return out


def epoch_indices(n_rows, first, count, seed, epoch, out=None, base=0, device="cuda"):
    """Positions first..first+count-1 of the keyed permutation (seed, epoch) of [0, n_rows) (int64, on the device)."""
    if out is None:
        out = torch.empty(count, dtype=torch.int64, device=device)
    N.check(N.lib().porl_epoch_indices(n_rows, first, count, seed, epoch, base, N.ptr(out), N.current_stream_ptr(out)),
            "porl_epoch_indices")
    return out


def prof_enable(on=True):
    N.check(N.lib().porl_prof_enable(int(on)), "porl_prof_enable")


def prof_read(max_entries=64):
    """[{name, launches, total_ms, flops, bytes}] — synchronises the device."""
    buf = (N.ProfEntry * max_entries)()
    n = N.lib().porl_prof_read(buf, max_entries)
    if n < 0:
        N.check(n, "porl_prof_read")
    return [dict(name=buf[i].name.decode(), launches=int(buf[i].launches), total_ms=float(buf[i].total_ms),
                 flops=float(buf[i].flops), bytes=float(buf[i].bytes)) for i in range(n)]


def tune_set(key, value):
    N.check(N.lib().porl_tune_set(key.encode(), int(value)), "porl_tune_set")
