"""Per-kernel time split of one CQL learn step (config 3) through the engine's HIP-event profiler."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from porl_amd import engine as E
from porl_amd.train.cql_trainer import CQLTrainer
from porl_amd.util.synth import make_discrete_transitions
dev = torch.device("cuda", 0)
S, A, B, N = 60, 10, 4096, 100_000
for fused in (1,):
    E.tune_set("qnet_fused", fused)
    torch.manual_seed(0)
    t = CQLTrainer(state_size=S, action_size=A, gamma=0.99, device=dev, batch_size=B)
    st, ac, rw, ns, dn = make_discrete_transitions(N, S, A, seed=0)
    rb = t.replay_buffer
    rb.states[:N], rb.actions[:N], rb.rewards[:N], rb.next_states[:N], rb.dones[:N] = st, ac, rw, ns, dn
    rb.size, rb.position = N, 0
    t.async_losses = True
    for _ in range(20): t.learn_device_sampled()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): t.learn_device_sampled()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    E.prof_enable(True)
    for _ in range(50): t.learn_device_sampled()
    prof = E.prof_read(); E.prof_enable(False)
    print(f"fused={fused}: {200/el:.0f} steps/s ({1e6*el/200:.1f} us/step)")
    for p in prof:
        if p["launches"]: print(f"   {p['name'][:60]:60s} {p['launches']/50:5.1f} launches/step  {1e3*p['total_ms']/50:7.1f} us/step")
E.tune_set("qnet_fused", 1)

# phase timeline of block 0 (shader clock, 100 MHz-based s_memtime counts at 2.1-2.4 GHz equivalents)
from porl_amd import _native as N
buf = torch.zeros(64, dtype=torch.int64, device=dev)
N.check(N.lib().porl_tune_set_ptr(b"qnet_stamps", N.ptr(buf)))
t.learn_device_sampled(); torch.cuda.synchronize()
N.check(N.lib().porl_tune_set_ptr(b"qnet_stamps", None))
raw = buf.cpu().numpy()
s = raw[:32]; s = s[s != 0]
d = (s[1:] - s[:-1])
print("   raw deltas:", [int(x) for x in d])
print("   total", int(s[-1] - s[0]))
g1 = raw[32:]; g1 = g1[g1 != 0]
if len(g1):
    print("   group 1 (relative to kernel entry):", [int(x - s[0]) for x in g1])
    print("   group 0 cumulative:", [int(x - s[0]) for x in s])
