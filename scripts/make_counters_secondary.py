"""Summaries of scripts/profile_counters_secondary.sh: per-launch HBM bytes (FETCH_SIZE / WRITE_SIZE passes, gfx950
correction as in make_profiles.py) and matrix-pipe activity (SQ pass) of the CQL step (config 3) and the SORL
encoder update (config 5), one JSON per workload."""
import glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd_pmc_table import table

P, OUT = sys.argv[1], sys.argv[2]
os.makedirs(OUT, exist_ok=True)
SIMDS = 256 * 4


def db(tag):
    return sorted(glob.glob(os.path.join(P, tag, "**", "*_results.db"), recursive=True))[0]


for wl, cmd in (("cql", "bench.py --workload cql --steps 40 --warmup 5"),
                ("enc", "bench.py --workload sorl_enc --batch 256 --steps 2 --warmup 1")):
    try:
        f, w, m = table(db(wl + "_fetch"), 2), table(db(wl + "_write"), 2), table(db(wl + "_mfma"), 2)
    except IndexError:
        print(wl, ": passes not found, skipped")
        continue
    out = {"_note": "rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES "
                    "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES — three separate runs, no trace) of `%s`.  Per-launch averages, "
                    "keys kernel@workgroups.  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 counts 64 B per 128-B "
                    "request for 16-B/lane reads); mfma_util = MFMA-busy cycles / 1024 SIMDs / busy shader cycles; "
                    "tflops = MOPS*512 / duration (f32 MFMA only: the bf16 products of the encoder's bf16 mode are not in "
                    "this counter)." % cmd, "kernels": {}}
    for k in m:
        e = m[k]
        busy = e["SQ_BUSY_CYCLES"] / max(1, e["SQ_BUSY_CYCLES_instances"])
        rec = {"launches": e["launches"], "avg_us": e["avg_ns"] / 1e3,
               "mfma_util": e["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / busy if busy else 0.0,
               "clock_ghz": busy / e["avg_ns"] if e["avg_ns"] else 0.0,
               "tflops": e["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / e["avg_ns"] / 1e3 if e["avg_ns"] else 0.0}
        if k in f and k in w:
            rec["hbm_bytes_per_launch"] = (2 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024
            rec["hbm_gbs"] = rec["hbm_bytes_per_launch"] / e["avg_ns"]
        out["kernels"][k] = rec
    json.dump(out, open(os.path.join(OUT, os.environ.get("PORL_ROUND", "r03") + "_counters_%s.json" % ("cql" if wl == "cql" else "sorl_enc")), "w"), indent=1)
    for k, r in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"])[:8]:
        print(f"{wl} {k[:60]:60s} {r['avg_us']:9.1f} us x{r['launches']:4d} mfma_util {r['mfma_util']:.3f} {r['tflops']:6.1f} TF  "
              f"hbm {r.get('hbm_bytes_per_launch', 0) / 1e6:9.2f} MB {r.get('hbm_gbs', 0):7.1f} GB/s")
