"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/porl_hip.h declares,
its host-side planner answers layout queries, and compute refuses to run without a HIP device."""
import os
import re
import subprocess
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden
from porl_amd import _native as N


def _header_functions():
    txt = open(os.path.join(REPO, "include", "porl_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(porl_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported():
    names = _header_functions()
    assert names == sorted(N.SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", N.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT\s+(porl_[a-z0-9_]+)", out))
    assert set(names) <= exported, set(names) - exported
    lib = N.lib()
    for n in names:
        assert hasattr(lib, n)
    assert lib.porl_abi_version() == N.ABI_VERSION


def test_code_object_targets_gfx950_only():
    out = subprocess.run(["strings", "-n", "6", N.LIB_PATH], capture_output=True, text=True).stdout
    archs = set(re.findall(r"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", out))
    assert archs == {"gfx950"}, archs


def test_layout_matches_module_parameters():
    from porl_amd.engine import IqlEngine
    from porl_amd.agent.value_functions import TwinV
    from porl_amd.agent.policy import GaussianPolicy
    for S, D, H, L, ln in [(60, 60, 1024, 2, False), (17, 5, 48, 3, False), (362, 2, 64, 1, False), (60, 60, 64, 2, True)]:
        eng = IqlEngine(S, D, H, L, layer_norm=ln, max_batch=8)
        vf, pol = TwinV(S, layer_norm=ln, hidden_dim=H, n_hidden=L), GaussianPolicy(S, D, hidden_dim=H, n_hidden=L)
        for mod, group, total in ((vf, 0, eng.n_vf), (pol, 1, eng.n_pol)):
            table = eng.tensor_table(group)
            shapes = [tuple(p.shape) for p in mod.parameters()]
            assert [s for _, s in table] == shapes
            end = 0
            for off, shape in table:
                assert off % 4 == 0 and off >= end            # 16-byte aligned, non-overlapping, ordered
                end = off + int(np.prod(shape))
            assert end <= total and total % 4 == 0
    # P_V = 1 113 089 per V-net, P_G = 1 173 624 (unpadded counts)
    eng = IqlEngine(60, 60, 1024, 2, max_batch=8)
    assert sum(int(np.prod(s)) for _, s in eng.tensor_table(0)) == 2 * 1113089
    assert sum(int(np.prod(s)) for _, s in eng.tensor_table(1)) == 1173624


def test_invalid_and_unsupported_configs():
    from porl_amd.engine import IqlEngine
    with pytest.raises(N.NativeError):
        IqlEngine(60, 60, 64, 0)                      # n_hidden < 1
    with pytest.raises(N.NativeError):
        IqlEngine(0, 60, 64, 2)
    with pytest.raises(N.NativeError, match="layer_norm"):
        IqlEngine(60, 60, 4096, 2, layer_norm=True)   # wider than the LayerNorm kernels: fails loudly, no fallback
    IqlEngine(60, 60, 1024, 2, layer_norm=True)


def test_no_cpu_fallback():
    from porl_amd.agent.por import POR
    args = SimpleNamespace(state_size=60, hidden_dim=64, n_hidden=2, layer_norm=False, action_size=2)
    agent = POR(args, 1000, 0.9, 10.0)                # device=cpu, as the reference's default
    x = torch.zeros(4, 60)
    with pytest.raises(N.NativeError, match="no CPU path"):
        agent.por_residual_update(x, x, torch.zeros(4), torch.zeros(4))
    with pytest.raises(N.NativeError):
        agent.vf.both(x)


def test_product_never_imports_oracle():
    import ast
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "porl_amd")):
        for f in files:
            if f.endswith(".py"):
                tree = ast.parse(open(os.path.join(root, f)).read())
                for node in ast.walk(tree):
                    mods = []
                    if isinstance(node, ast.Import):
                        mods = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom) and node.module:
                        mods = [node.module]
                    bad += [(f, m) for m in mods if m.split(".")[0] == "oracle"]
    assert not bad, bad


def test_host_planner_under_sanitizers():
    """the host half of csrc/porl_api.hip (validation, layout, planning) built with
    -fsanitize=address,undefined and driven over the rejected-argument paths and the pure-host queries of every engine
    (tests/helpers/abi_reject.cpp: no kernel launch, so no GPU is needed).  The driver returns 0 only when every bad call
    came back as a clean error with a message; the sanitizers abort the process on any finding."""
    import subprocess
    from porl_amd.build import build_sanitized
    driver = build_sanitized(verbose=False)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([driver], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "0 unexpected" in r.stdout and "checks" in r.stdout
    n = int(r.stdout.split("abi_reject:")[1].split("checks")[0])
    assert n >= 100
