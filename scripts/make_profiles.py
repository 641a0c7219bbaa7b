"""Turn the rocprofv3 outputs of scripts/profile_round.sh (gpurun_out/<round>/prof/*, round = PORL_ROUND, default r03) into the committed summaries under
profiles/: kernel-stats CSVs, the per-launch step table, HBM bytes per launch (FETCH_SIZE / WRITE_SIZE passes) and
MFMA utilisation (SQ_* pass)."""
import collections, csv, json, os, sqlite3, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RD = os.environ.get("PORL_ROUND", "r03")
P = os.path.join(REPO, "gpurun_out", RD, "prof")
OUT = os.environ.get("PORL_PROFILES_OUT", os.path.join(REPO, "profiles"))    # on the GPU box: a directory under gpurun_out/
os.makedirs(OUT, exist_ok=True)
sys.path.insert(0, os.path.join(REPO, "scripts"))
from rocpd_pmc_table import table


def stats_csv(db, out):
    c = sqlite3.connect(db).cursor()
    st = collections.OrderedDict()
    for n, s, e in c.execute("select name, start, end from kernels order by start"):
        st.setdefault(n, []).append(e - s)
    total = sum(sum(v) for v in st.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([n, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 2), min(v), max(v)])


for tag, name in (("por_serial", RD + "_kernel_stats.csv"), ("por_pipelined", RD + "_kernel_stats_pipelined.csv"),
                  ("cql", RD + "_kernel_stats_cql.csv"), ("enc_fp32", RD + "_kernel_stats_sorl_enc.csv"),
                  ("enc_bf16", RD + "_kernel_stats_sorl_enc_bf16.csv"), ("enc_bf16_84", RD + "_kernel_stats_sorl_enc_bf16_84x84.csv")):
    if not os.path.exists(os.path.join(P, tag, "t_results.db")):
        continue
    stats_csv(os.path.join(P, tag, "t_results.db"), os.path.join(OUT, name))
    js = os.path.join(P, tag + ".json")
    if os.path.exists(js):
        with open(js) as f, open(os.path.join(OUT, name.replace("kernel_stats", "bench_under_rocprof").replace(".csv", ".json")), "w") as g:
            g.write(f.read())
with open(os.path.join(OUT, RD + "_step_table.txt"), "w") as f:
    f.write("rocprofv3 --kernel-trace of `bench.py --steps 200 --warmup 20 --no-pipeline` (every update back to back on one "
            "stream, so each kernel has the chip to itself): average duration of the k-th launch of the update over the last "
            "150 updates.\n\n")
    f.write(subprocess.run([sys.executable, os.path.join(REPO, "scripts", "rocpd_step_table.py"),
                            os.path.join(P, "por_serial", "t_results.db"), "15", "150"], capture_output=True, text=True).stdout)

fetch, write = table(os.path.join(P, "pmc_fetch", "t_results.db"), 2), table(os.path.join(P, "pmc_write", "t_results.db"), 2)
hbm = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 40 --warmup 5 --no-pipeline`; "
                "per-launch averages over all launches of a kernel, counter instances summed.  bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: "
                "on gfx950 FETCH_SIZE counts 64 B per 128-B request for 16-B/lane reads (MI355X_MICROARCH.md, HBM), so the read "
                "side is doubled.", "kernels": {}}
hbm["_note"] += "  Keys are kernel@workgroups (launch shapes kept apart); a plain kernel name repeats the entry of its largest launch."
for k in fetch:
    if k in write:
        hbm["kernels"][k] = {
            "fetch_size_kb": fetch[k]["FETCH_SIZE"], "write_size_kb": write[k]["WRITE_SIZE"], "launches": fetch[k]["launches"],
            "hbm_bytes_per_launch": (2 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"]) * 1024}
for k in list(hbm["kernels"]):                     # plain name -> the shape with the most bytes (the dominant launch)
    base = k.split("@")[0]
    if base not in hbm["kernels"] or hbm["kernels"][k]["hbm_bytes_per_launch"] > hbm["kernels"][base]["hbm_bytes_per_launch"]:
        hbm["kernels"][base] = dict(hbm["kernels"][k], shape=k)
json.dump(hbm, open(os.path.join(OUT, RD + "_hbm_traffic.json"), "w"), indent=1)

m = table(os.path.join(P, "pmc_mfma", "t_results.db"), 2)
SIMDS = 256 * 4
mf = {"_note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES (one pass, no trace) "
               "of `bench.py --steps 40 --warmup 5 --no-pipeline`; per-launch averages, the 32 counter instances summed.  "
               "SQ_INSTS_VALU_MFMA_MOPS_F32 counts 512 FLOP per unit (2^24 units = the 8.59 GFLOP of the 4 x 1024^3 launch); "
               "SQ_VALU_MFMA_BUSY_CYCLES = 64 cycles per v_mfma_f32_32x32x2_f32, summed over the 1024 SIMDs.  "
               "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 instances): the fraction of the kernel's "
               "busy shader cycles in which a SIMD's matrix pipe is occupied; clock_ghz = busy cycles / kernel duration; "
               "tflops = MOPS * 512 / duration (counters add ~7 % to the duration).", "kernels": {}}
for k, e in m.items():
    busy = e["SQ_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES_instances"]
    mf["kernels"][k] = {"launches": e["launches"], "avg_us": e["avg_ns"] / 1e3,
                        "SQ_VALU_MFMA_BUSY_CYCLES": e["SQ_VALU_MFMA_BUSY_CYCLES"], "SQ_BUSY_CYCLES": e["SQ_BUSY_CYCLES"],
                        "SQ_INSTS_VALU_MFMA_MOPS_F32": e["SQ_INSTS_VALU_MFMA_MOPS_F32"], "SQ_WAVE_CYCLES": e["SQ_WAVE_CYCLES"],
                        "mfma_util": e["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / busy if busy else 0.0,
                        "clock_ghz": busy / e["avg_ns"],
                        "tflops": e["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / e["avg_ns"] / 1e3}
json.dump(mf, open(os.path.join(OUT, RD + "_mfma_util.json"), "w"), indent=1)
for k, e in mf["kernels"].items():
    print(f"{k[:52]:52s} {e['avg_us']:8.1f} us  mfma_util {e['mfma_util']:.3f}  clock {e['clock_ghz']:.2f} GHz  {e['tflops']:.1f} TF")
print(open(os.path.join(OUT, RD + "_step_table.txt")).read())
