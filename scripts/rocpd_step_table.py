"""Per-launch table of one update step from a rocprofv3 (rocpd sqlite) kernel trace: average duration of the k-th
launch of the step over the last N steps, the gaps between launches, and a kernel-stats CSV like `--stats` writes."""
import collections, csv, sqlite3, sys

db_path, per_step = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 150
out_csv = sys.argv[4] if len(sys.argv) > 4 else None
c = sqlite3.connect(db_path).cursor()
rows = [r for r in c.execute("select name, start, end from kernels order by start") if "at::native" not in r[0] and "rocclr" not in r[0]]
seq = rows[-per_step * steps:]
agg = collections.OrderedDict()
for i, (n, s, e) in enumerate(seq):
    agg.setdefault((i % per_step, n.split("(")[0].replace("void porl::", "").replace("porl::", "")), []).append((e - s) / 1e3)
tot = 0.0
for (k, n), v in agg.items():
    print(f"{k:3d} {n[:70]:70s} {sum(v) / len(v):8.2f} us  x{len(v)}")
    tot += sum(v) / len(v)
gaps = [(seq[i + 1][1] - seq[i][2]) / 1e3 for i in range(len(seq) - 1)]
span = (seq[-1][2] - seq[0][1]) / 1e3 / steps
print(f"sum of kernel time per step {tot:.1f} us; mean gap between launches {sum(gaps) / len(gaps):.2f} us; wall per step {span:.1f} us")
if out_csv:
    st = collections.OrderedDict()
    for n, s, e in rows:
        st.setdefault(n, []).append(e - s)
    total = sum(sum(v) for v in st.values())
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([n, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 2), min(v), max(v)])
