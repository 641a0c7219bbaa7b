"""Shared host logic of the POR and SORL agents: parameter adoption into the engine's flat groups,
optimizer / scheduler facades with torch-compatible state, and the (optionally data-parallel) update.
"""
from __future__ import annotations

import math
import warnings

import torch
import torch.nn as nn

from ..engine import IqlEngine
from ..parallel import GradExchange



# cross-stream ordering of the pipelined update: "signal" = counters in signal memory (engine.py: 3 380 vs 3 300 updates/s),
# "event" = event record / stream-wait-event pairs (A/B, and the fallback where wait-value operations are missing)
_PIPE_SYNC = __import__("os").environ.get("PORL_PIPE_SYNC", "signal2")
# The whole pipelined update from one native call (porl_iql_update_pipelined): "1" always, "0" never, default "auto" =
# only for small networks, where the host's issue rate is the bound (measured, updates/s, phase calls -> one call:
# H=256 B=256 9 470 -> 10 990; H=256 B=1024 8 160 -> 8 060; H=1024 B=1024 3 365 -> 3 315 — at GPU-bound sizes the
# denser issue order shifts the two streams against each other and costs ~1 %).
_PIPE_ONECALL = __import__("os").environ.get("PORL_PIPE_ONECALL", "auto")
_DP_POLICY_GROUP = __import__("os").environ.get("PORL_DP_POLICY_GROUP", "1") != "0"
_ONECALL_MAX_WORK = 32 << 20        # batch * hidden_dim^2 below which the update is host-bound

class ArenaAdam:
    """torch.optim.Adam look-alike over one flat parameter group of the engine.

    Arithmetic follows torch's single-tensor Adam; the sweep itself is the HIP
    kernel `adam_ema_kernel`.  `state_dict()` / `load_state_dict()` use torch.optim.Adam's format so
    optimizer checkpoints interchange with the reference.
    """

    def __init__(self, agent, group, named_params, lr):
        self._agent, self._group = agent, group
        self._names = [n for n, _ in named_params]
        self._params = [p for _, p in named_params]
        self.step_count = 0
        self.defaults = dict(lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False,
                             maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        self.param_groups = [dict(self.defaults, params=self._params, initial_lr=lr)]

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def _moments(self):
        eng = self._agent._engine
        flat_m = eng.adam_m_vf if self._group == IqlEngine.GROUP_VF else eng.adam_m_pol
        flat_v = eng.adam_v_vf if self._group == IqlEngine.GROUP_VF else eng.adam_v_pol
        table = eng.tensor_table(self._group)
        return IqlEngine.views(flat_m, table), IqlEngine.views(flat_v, table)

    def zero_grad(self, set_to_none=True):
        pass  # gradients are overwritten by every backward pass of the engine

    def step(self):
        raise RuntimeError("ArenaAdam steps inside the fused update (por_residual_update / update)")

    sharded = False     # set by the reduce-scatter exchange: each rank holds current moments for its slice only

    def _gather_moments(self):
        if self.sharded:
            eng, ex = self._agent._engine, self._agent._exchange
            _, _, m, v, _ = eng.group(self._group)
            ex.all_gather_(m)
            ex.all_gather_(v)

    def _unshard(self):
        """Before a whole-group Adam on every rank (all_reduce exchange): moments of the other ranks' slices are stale
        after reduce-scatter updates, so gather them once (collective: every rank takes the same branch because
        `grad_exchange` is per-agent configuration, identical on all ranks)."""
        if self.sharded:
            self._gather_moments()
            self.sharded = False

    def consolidate_state(self):
        """Data-parallel checkpointing: call on EVERY rank; afterwards `state_dict()` is purely local, so the usual
        `if rank == 0: torch.save(opt.state_dict())` cannot deadlock.  The next sharded update marks the state as
        sharded again."""
        self._agent.flush()
        self._unshard()

    def state_dict(self):
        """torch.optim.Adam's format.  DATA-PARALLEL NOTE: after updates in the default `grad_exchange=
        "reduce_scatter"` mode this call all-gathers the moments and is therefore a COLLECTIVE — every rank must make
        it (or every rank calls `consolidate_state()` first and rank 0 alone calls `state_dict()`)."""
        self._agent.flush()
        self._gather_moments()
        ms, vs = self._moments()
        state = {}
        if self.step_count > 0:
            for i, (m, v) in enumerate(zip(ms, vs)):
                state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=m.clone(), exp_avg_sq=v.clone())
        groups = [{k: v for k, v in self.param_groups[0].items() if k != "params"}]
        groups[0]["params"] = list(range(len(self._params)))
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        self._agent.flush()
        ms, vs = self._moments()
        steps = set()
        with torch.no_grad():
            for i, st in sd["state"].items():
                ms[int(i)].copy_(st["exp_avg"])
                vs[int(i)].copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter Adam step counts differ; the flat sweep needs one counter")
        self.step_count = steps.pop() if steps else 0
        self.sharded = False               # every slice was just written on this rank (all ranks load the same file)
        g = sd["param_groups"][0]
        for k in ("lr", "betas", "eps", "initial_lr"):
            if k in g:
                self.param_groups[0][k] = g[k]


class CosineSchedule:
    """CosineAnnealingLR(T_max, eta_min=0) in closed form: equals torch's
    recursive form to 2e-19, including the oscillation past T_max that the reference's runs exhibit."""

    def __init__(self, optimizer: ArenaAdam, T_max):
        self.optimizer, self.T_max, self.last_epoch = optimizer, T_max, 0
        self.base_lrs = [optimizer.param_groups[0]["initial_lr"]]

    def _lr(self, t):
        return self.base_lrs[0] * (1.0 + math.cos(math.pi * t / self.T_max)) / 2.0

    def step(self):
        self.last_epoch += 1
        self.optimizer.param_groups[0]["lr"] = self._lr(self.last_epoch)

    def get_last_lr(self):
        return [self.optimizer.param_groups[0]["lr"]]

    def state_dict(self):
        return dict(T_max=self.T_max, last_epoch=self.last_epoch, base_lrs=list(self.base_lrs))

    def load_state_dict(self, sd):
        self.T_max, self.last_epoch = sd["T_max"], sd["last_epoch"]
        self.base_lrs = list(sd["base_lrs"])
        self.optimizer.param_groups[0]["lr"] = self._lr(self.last_epoch)


class IqlAgentBase(nn.Module):
    """Owns the engine; subclasses set the module attribute names the reference uses."""

    _warned_nll = False
    pipeline = True     # async_losses mode only: overlap the policy phase with the next update's value phase
    # data-parallel exchange: "reduce_scatter" = reduce-scatter of the flat gradient group -> Adam on this rank's 1/N
    # slice (N-fold less optimizer traffic) -> all-gather of the parameters (+ a local Polyak sweep for the target);
    # "all_reduce" = SUM all-reduce of the whole group, Adam everywhere.  Same arithmetic per element.
    grad_exchange = "reduce_scatter"
    # pipelined data-parallel mode: the policy group's collectives on a process group (RCCL communicator) of their own
    # (see _exchange_for); False keeps ONE communicator — the conservative choice (PORL_DP_POLICY_GROUP=0 sets it too)
    dp_policy_group = _DP_POLICY_GROUP

    def _setup_engine(self, vf: nn.Module, v_target: nn.Module, policy: nn.Module, *, obs_dim, pol_out_dim,
                      hidden_dim, n_hidden, layer_norm, pol_tanh, weight_mode, device, max_batch):
        self._engine = IqlEngine(obs_dim, pol_out_dim, hidden_dim, n_hidden, layer_norm, pol_tanh, weight_mode,
                                 max_batch, device)
        self._mods = (vf, v_target, policy)
        self._adopt(copy_from_modules=True)
        self._exchange = GradExchange()
        self._gslice = {}
        self.async_losses = False     # True: return a (3,) device VIEW [v_loss, g_loss, min_nll], no host sync

    def _adopt(self, copy_from_modules=False):
        """Point every nn.Parameter at its view in the engine's flat tensors."""
        eng = self._engine
        vf, v_target, policy = self._mods
        pairs = ((vf, eng.params_vf, IqlEngine.GROUP_VF, "vf"),
                 (v_target, eng.params_tgt, IqlEngine.GROUP_VF, "target"),
                 (policy, eng.params_pol, IqlEngine.GROUP_POL, "policy"))
        with torch.no_grad():
            for mod, flat, group, role in pairs:
                views = IqlEngine.views(flat, eng.tensor_table(group))
                params = list(mod.parameters())
                if len(params) != len(views):
                    raise RuntimeError(f"{role}: {len(params)} parameters vs {len(views)} engine tensors")
                for p, v in zip(params, views):
                    if tuple(p.shape) != tuple(v.shape):
                        raise RuntimeError(f"{role}: parameter shape {tuple(p.shape)} vs engine {tuple(v.shape)}")
                    if copy_from_modules:
                        v.copy_(p)
                    p.data = v
                mod._engine, mod._engine_role, mod._private = eng, role, False
        for p in v_target.parameters():
            p.requires_grad_(False)

    def _apply(self, fn, recurse=True):
        # .to()/.cuda()/.cpu(): move the flat tensors, then re-create the parameter views
        self.flush()
        probe = fn(torch.empty(0, dtype=torch.float32, device=self._engine.device))
        if probe.dtype != torch.float32:
            raise RuntimeError("porl_amd agents are fp32 only")
        self._engine.to(probe.device)
        self.device = probe.device
        self._adopt()
        backbone = getattr(self, "backbone", None)
        if backbone is not None:
            backbone._apply(fn)
        return self

    # ---------------------------------------------------------------------------------------------
    def _hyper(self, batch, v_opt: ArenaAdam, p_opt: ArenaAdam):
        world = self._exchange.world_size
        return self._engine.hyper(
            tau=self.tau, discount=self.discount, alpha=self.alpha, ema_beta=self.beta,
            inv_batch=1.0 / (batch * world), value_lr=v_opt.lr, policy_lr=p_opt.lr,
            value_step=v_opt.step_count, policy_step=p_opt.step_count,
            adam_beta1=v_opt.param_groups[0]["betas"][0], adam_beta2=v_opt.param_groups[0]["betas"][1],
            adam_eps=v_opt.param_groups[0]["eps"])

    def _full_update(self, obs, next_obs, rew, term, pol_target, v_opt, p_opt, sched, replay=None, batch=None):
        """One update = value phase (load, forward, backward, [exchange], Adam + target EMA) then policy phase
        (second twin forward, weights, NLL, backward, [exchange], Adam), reference por.py:81-110.

        async_losses=False (the reference's behaviour, losses returned as floats): everything on the current stream.
        async_losses=True: PIPELINED — the policy phase of update t is issued on the engine's side stream and the
        call returns; update t+1's value phase then runs beside it on the caller's stream.  The only data the two
        share are the value parameters (policy phase t reads the vf of update t), so two events order
        value Adam(t) -> policy phase(t) -> value Adam(t+1); minibatch staging is double-buffered in the engine.
        Same arithmetic, same results; `flush()` (called by everything that reads the agent) joins the streams.  The
        returned statistics view is complete only after `flush()` (or a device synchronisation): g_loss and min NLL of
        the last update are written by the side stream.
        With a data-parallel group the two gradient exchanges ride on their phase's stream, so the policy-group
        all-reduce and most of the value-group one overlap with the other phase's kernels."""
        eng, ex = self._engine, self._exchange
        world = ex.world_size
        dp = ex.active                 # world > 1, or a one-rank group with the exchange forced on (parallel.GradExchange)
        pipelined = self.async_losses and self.pipeline
        if not pipelined:
            eng.join()
        eng._ensure_bound()
        eng.set_mode(((IqlEngine.MODE_TWO_SLOTS | IqlEngine.MODE_SHORT_BLOCKS) if pipelined else 0) |
                     (IqlEngine.MODE_FOLD_COMBINE if not dp else 0))
        use_sig = pipelined and _PIPE_SYNC in ("signal", "signal2") and eng.signals() is not None
        onecall = _PIPE_ONECALL is True or _PIPE_ONECALL == "1" or (
            _PIPE_ONECALL == "auto" and (batch or 0) * eng.cfg.hidden_dim ** 2 <= _ONECALL_MAX_WORK)
        if use_sig and not dp and replay is not None and onecall:
            # the whole update from one native call (csrc: porl_iql_update_pipelined): the same operations on the same
            # two streams in the same order as the phase calls below, without ~12 trips through ctypes per update
            eng._seq += 1
            seq = eng._seq
            v_opt.step_count += 1
            p_opt.step_count += 1
            hp = self._hyper(batch, v_opt, p_opt)
            vr = eng._values_read
            try:
                eng.update_pipelined(hp, replay, batch, seq,
                                     wait_policy_seq=max(0, seq - eng.SLOTS) if _PIPE_SYNC == "signal" else 0,
                                     wait_fwd_seq=vr[1] if isinstance(vr, tuple) else 0,
                                     write_policy=_PIPE_SYNC == "signal")
            except Exception:          # rejected arguments (e.g. batch > max_batch): nothing was launched, undo the counters
                eng._seq -= 1
                v_opt.step_count -= 1
                p_opt.step_count -= 1
                raise
            replay.draws += 1
            eng._values_read, eng._policy_done = ("sig", seq), True
            sched.step()
            return self._losses()
        if use_sig:
            main = torch.cuda.current_stream(eng.device)
            eng._seq += 1
            seq = eng._seq
            # The staging slot loaded next was last read by the policy phase SLOTS = 3 updates ago.  No wait is needed for
            # it: this stream has already waited (before its previous value Adam) for the forward half of policy phase
            # seq - 2, which the in-order side stream only starts after policy phase seq - 3 has finished.  PORL_PIPE_SYNC=
            # signal adds the explicit wait and the counter write it pairs with: a stream operation holds its queue for
            # ~5 us plus a gap, and with round 3's kernels the two extra ones cost 2.5-3 % (sustained 3 256-3 291 against
            # 3 342-3 386 updates/s, two boxes, two runs each, gpurun_out/r03/sig2*; round 2 had measured +0.7 % FOR them,
            # which is why they were there).
            if _PIPE_SYNC == "signal":
                eng.wait_signal(eng.SIG_POLICY, seq - eng.SLOTS, main)
        elif pipelined:
            eng.wait_slot_free()               # the policy phase SLOTS updates ago used the staging slot loaded next
        if replay is not None:
            B = eng.load_batch_sampled(replay.rows, batch, replay.seed, replay.draws, replay.act_dim,
                                       self._engine.cfg.weight_mode == 1)
            replay.draws += 1
        else:
            B = eng.load_batch(obs, next_obs, rew, term, pol_target)
        v_opt.step_count += 1
        p_opt.step_count += 1
        hp = self._hyper(B, v_opt, p_opt)
        if not dp and not pipelined:
            eng.step(hp)
            sched.step()
            return self._losses()
        # ---- value phase (current stream) -----------------------------------------------------------------------
        # replay sharded across ranks: each rank's gradients carry 1/B_global, so SUM == global mean
        eng.value_backward(hp)
        if dp and self._sharded():
            self._sharded_apply(IqlEngine.GROUP_VF, hp, v_opt)
        else:
            if dp:
                v_opt._unshard()
                ex.allreduce_sum_(eng.grads_vf)
            eng.wait_values_read()             # the PREVIOUS update's policy phase has read the old value nets
            eng.value_apply(hp)
        # ---- policy phase -------------------------------------------------------------------------------------------
        if pipelined:
            main, side = torch.cuda.current_stream(eng.device), eng.side_stream()
            k = (v_opt.step_count % eng.SLOTS) * 3
            ev_v, ev_f, ev_p = eng.event(k), eng.event(k + 1), eng.event(k + 2)
            if use_sig:
                eng.signal(eng.SIG_VALUE, seq, main)
            else:
                ev_v.record(main)
            with torch.cuda.stream(side):
                if use_sig:
                    eng.wait_signal(eng.SIG_VALUE, seq, side)
                else:
                    side.wait_event(ev_v)
                eng.policy_forward(hp)                    # every read of the value nets the policy phase makes
                if use_sig:
                    eng.signal(eng.SIG_FWD, seq, side)
                else:
                    ev_f.record(side)
                eng.policy_backward(hp)
                if dp and self._sharded():                # loss statistics stay per-rank shares in this mode
                    self._sharded_apply(IqlEngine.GROUP_POL, hp, p_opt, self._exchange_for(IqlEngine.GROUP_POL, True))
                else:
                    if dp:
                        p_opt._unshard()
                        self._exchange_for(IqlEngine.GROUP_POL, True).allreduce_sum_(eng.grads_pol)
                    eng.policy_apply(hp)
                if _PIPE_SYNC == "signal" and use_sig:
                    eng.signal(eng.SIG_POLICY, seq, side)
                if not use_sig:                               # (signal mode: join() records its own event when needed)
                    ev_p.record(side)
            # the next value Adam waits for ev_f only; readers of the agent (flush) wait for ev_p
            eng._values_read, eng._policy_done = (("sig", seq) if use_sig else ev_f), (True if use_sig else ev_p)
            eng._slot_users = eng._slot_users[1:] + [None if use_sig else ev_p]
        else:
            eng.policy_backward(hp)
            if self._sharded():
                self._sharded_apply(IqlEngine.GROUP_POL, hp, p_opt)
            else:
                p_opt._unshard()
                ex.allreduce_sum_(eng.grads_pol)
                eng.policy_apply(hp)
            if not self.async_losses:              # async mode: statistics stay per-rank SHARES (no per-update collective);
                ex.allreduce_stats_(eng.stats)     # the caller reduces the history once (bench.py)
        sched.step()
        return self._losses()

    def _sharded(self):
        eng, ex = self._engine, self._exchange
        return self.grad_exchange == "reduce_scatter" and ex.can_shard(eng.grads_vf) and ex.can_shard(eng.grads_pol)

    def _exchange_for(self, group, pipelined):
        """The exchange object a phase's collectives go through.  torch.distributed runs all collectives of ONE process
        group (one RCCL communicator) on one internal stream in issue order, so with a single group the policy
        exchange of update t (issued on the side stream, ready only when that phase's backward is done) would sit in
        front of the value exchange of update t+1 and serialise the two streams again.  In pipelined mode the policy
        group therefore gets a process group of its own (same ranks: `dist.new_group()`, created on first use by every
        rank at the same point of the program); PORL_DP_POLICY_GROUP=0 keeps the single group."""
        ex = self._exchange
        if group != IqlEngine.GROUP_POL or not pipelined or not ex.active or not self.dp_policy_group:
            return ex
        if getattr(self, "_exchange_pol", None) is None:
            import torch.distributed as dist
            self._exchange_pol = GradExchange(dist.new_group(), force=ex.force)
            # build the new group's communicator now, with nothing else of this agent in flight behind it: one tiny
            # collective and a device synchronisation (once per agent; every rank reaches this point in its first
            # pipelined update)
            probe = torch.zeros(1, dtype=torch.float32, device=self._engine.device)
            dist.all_reduce(probe, group=self._exchange_pol.group)
            torch.cuda.synchronize(self._engine.device)
        return self._exchange_pol

    def _sharded_apply(self, group, hp, opt, ex=None):
        """reduce-scatter(SUM) of the group's gradients, torch-exact Adam on this rank's slice only,
        all-gather of the updated parameters; the target network is then swept locally (Polyak) from the gathered
        parameters.  Moments of the other slices are not kept here: ArenaAdam.state_dict() gathers them."""
        from .. import engine as E
        eng = self._engine
        ex = ex or self._exchange
        p, g, m, v, tgt = eng.group(group)
        gs = self._gslice.get(group)
        if gs is None or gs.numel() != p.numel() // ex.world_size:
            gs = self._gslice[group] = torch.empty(p.numel() // ex.world_size, dtype=torch.float32, device=p.device)
        ex.reduce_scatter_sum(g, gs)
        if group == IqlEngine.GROUP_VF:
            eng.wait_values_read()             # the PREVIOUS update's policy phase has read the old value nets
        b1, b2 = opt.param_groups[0]["betas"]
        E.adam_ema(ex.slice_of(p), gs, ex.slice_of(m), ex.slice_of(v), None, opt.lr, opt.step_count, b1, b2,
                   opt.param_groups[0]["eps"], 0.0)
        ex.all_gather_(p)
        if tgt is not None:
            E.ema(tgt, p, hp.ema_beta)
        opt.sharded = True

    def flush(self):
        """Join an outstanding policy phase (pipelined mode) into the current stream; a no-op otherwise.  Called by
        everything that reads or writes the agent's state: state_dict / load_state_dict, the optimizers' state,
        module forwards, vf_update, .to()."""
        self._engine.join()

    def state_dict(self, *args, **kwargs):
        self.flush()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.flush()
        return super().load_state_dict(*args, **kwargs)

    def _value_update(self, obs, next_obs, rew, term, v_opt):
        eng, ex = self._engine, self._exchange
        eng.join()
        eng._ensure_bound()
        eng.set_mode(IqlEngine.MODE_FOLD_COMBINE if not ex.active else 0)
        B = eng.load_batch(obs, next_obs, rew, term, None)
        v_opt.step_count += 1
        hp = self._hyper(B, v_opt, v_opt)
        eng.value_backward(hp)
        if ex.active and self._sharded():
            # same exchange as the full update: after a reduce-scatter update a rank holds current Adam moments for its
            # own slice only, so a whole-group Adam here would use stale moments everywhere else
            self._sharded_apply(IqlEngine.GROUP_VF, hp, v_opt)
            ex.allreduce_stats_(eng.stats)
            return
        if ex.active:
            v_opt._unshard()                   # (all_reduce mode after sharded updates: make every slice current first)
            ex.allreduce_sum_(eng.grads_vf)
            ex.allreduce_stats_(eng.stats)
        eng.value_apply(hp)

    def _losses(self):
        if self.async_losses:
            # a VIEW of the engine's statistics buffer: no copy, no sync; the next update overwrites it
            # unless the caller moved the buffer with engine.set_stats(...)
            return self._engine.stats[:3]
        v_loss, g_loss, min_nlp = self._engine.stats[:3].tolist()     # the one host sync of the update
        if math.isnan(g_loss) or math.isnan(v_loss):
            # the reference fails earlier (MultivariateNormal validate_args -> ValueError)
            raise ValueError("NaN loss: non-finite values in the minibatch or the parameters")
        if min_nlp <= 0 and not IqlAgentBase._warned_nll:
            IqlAgentBase._warned_nll = True
            warnings.warn("per-sample NLL <= 0 in this batch (the reference drops into pdb here, "
                          "agent/por.py:104-105); continuing", RuntimeWarning)
        self.last_min_nll = min_nlp
        return v_loss, g_loss
