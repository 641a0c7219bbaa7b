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


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
