"""Build libporl_hip.so for gfx950 in-tree:  python -m porl_amd.build

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "porl_api.hip")
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))) + \
       [os.path.join(os.path.dirname(HERE), "include", "porl_hip.h")]
OUT = os.path.join(HERE, "lib", "libporl_hip.so")


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


HASH = OUT + ".srchash"


def source_hash():
    """sha256 over the library's sources (csrc/* and the header), in name order."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def up_to_date():
    """The binary is git-ignored and travels by snapshot, so staleness is decided by CONTENT: the hash of the sources it
    was built from is kept beside it (modification times do not survive a snapshot reliably)."""
    try:
        return os.path.exists(OUT) and open(HASH).read().strip() == source_hash()
    except OSError:
        return False


def build(force=False, verbose=True):
    sh = source_hash()
    if not force and up_to_date():
        if verbose:
            print(f"libporl_hip.so is current (sources sha256 {sh[:16]})", flush=True)
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(HASH, "w") as f:
        f.write(sh + "\n")
    if verbose:
        print(f"built from sources sha256 {sh[:16]}", flush=True)
    return OUT


SAN_OUT = os.path.join(HERE, "lib", "libporl_hip_host_san.so")
SAN_DRIVER_SRC = os.path.join(os.path.dirname(HERE), "tests", "helpers", "abi_reject.cpp")
SAN_DRIVER = os.path.join(HERE, "lib", "abi_reject_san")
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


def build_sanitized(force=False, verbose=True):
    """The same translation unit with AddressSanitizer + UBSan on the HOST half (planner, validation, layout code; the
    device half is compiled as usual and never runs), plus the driver that walks the rejected-argument paths
    (tests/helpers/abi_reject.cpp).  CPU-only check: GPU sanitizers are not available on this pool.  Returns the
    driver's path; cached by source hash like the product library."""
    import hashlib
    h = hashlib.sha256((source_hash() + open(SAN_DRIVER_SRC).read()).encode()).hexdigest()
    tag = SAN_OUT + ".srchash"
    try:
        if not force and os.path.exists(SAN_OUT) and os.path.exists(SAN_DRIVER) and open(tag).read().strip() == h:
            return SAN_DRIVER
    except OSError:
        pass
    os.makedirs(os.path.dirname(SAN_OUT), exist_ok=True)
    clangxx = os.path.join(os.path.dirname(os.path.dirname(hipcc())), "lib", "llvm", "bin", "clang++")
    cmds = [[hipcc(), "--offload-arch=gfx950", "-fno-gpu-sanitize", *SAN_FLAGS, "-O1", "-std=c++17", "-shared", "-fPIC",
             "-o", SAN_OUT, SRC],
            [clangxx if os.path.exists(clangxx) else "clang++", "-std=c++17", "-O1", "-g", *SAN_FLAGS, "-o", SAN_DRIVER,
             SAN_DRIVER_SRC, SAN_OUT, "-Wl,-rpath," + os.path.dirname(SAN_OUT)]]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL if not verbose else None)
    with open(tag, "w") as f:
        f.write(h + "\n")
    return SAN_DRIVER


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
    if "--sanitized" in sys.argv:
        print(build_sanitized(force="--force" in sys.argv))
