"""Two-stream timeline of the pipelined update from a rocprofv3 (rocpd sqlite) kernel trace.

Launches are split by queue; on every queue the k-th launch of an update is averaged over the last N updates, with
start / end relative to the start of that update's first value-stream launch (`sampled_batch_kernel`).  Shows where a
stream waits (gap before a launch) and how far launches stretch beside the other stream's work.

usage: rocpd_timeline.py t_results.db [updates=100]"""
import collections, sqlite3, sys

db, nupd = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 100
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols), None)
rows = list(con.execute(f"select name, start, end, {qcol or '0'} from kernels order by start"))
rows = [r for r in rows if "at::native" not in r[0]]


def short(n):
    return n.split("(")[0].replace("void porl::", "").replace("porl::", "").replace("__amd_rocclr_", "")[:44]


anchors = [r[1] for r in rows if "sampled_batch" in r[0]]
anchors = anchors[-(nupd + 2):-1]
period = (anchors[-1] - anchors[0]) / (len(anchors) - 1) / 1e3
print(f"queues by column {qcol}; {len(anchors) - 1} updates, period {period:.1f} us")
by_q = collections.defaultdict(list)
for r in rows:
    by_q[r[3]].append(r)
for q, rs in sorted(by_q.items(), key=lambda kv: -len(kv[1]))[:2]:
    # position of a launch inside its update on this queue: count launches since the queue's own marker kernel
    names = [short(r[0]) for r in rs]
    marker = "sampled_batch_kernel" if any("sampled_batch" in n for n in names) else None
    if marker is None:
        # side stream: the update starts with the launch that follows an adam_ema_kernel
        idx = [i + 1 for i, n in enumerate(names[:-1]) if n.startswith("adam_ema")]
    else:
        idx = [i for i, n in enumerate(names) if n == marker]
    idx = idx[-(nupd + 1):-1]
    agg = collections.OrderedDict()
    for a, b in zip(idx[:-1], idx[1:]):
        t0 = rs[a][1]
        # anchor = the latest value-stream update start not after this launch sequence's start
        base = max((x for x in anchors if x <= t0), default=t0) if marker is None else t0
        for k in range(a, b):
            agg.setdefault((k - a, names[k]), []).append(((rs[k][1] - base) / 1e3, (rs[k][2] - base) / 1e3,
                                                          (rs[k][1] - rs[k - 1][2]) / 1e3 if k > 0 else 0.0))
    print(f"\nqueue {q}: {len(rs)} launches")
    print(f"{'k':>3} {'kernel':44s} {'start':>8} {'end':>8} {'dur':>7} {'gap before':>10}")
    for (k, n), v in agg.items():
        if len(v) < len(idx) // 2:
            continue
        s = sum(x[0] for x in v) / len(v); e = sum(x[1] for x in v) / len(v); g = sum(x[2] for x in v) / len(v)
        print(f"{k:3d} {n:44s} {s:8.1f} {e:8.1f} {e - s:7.1f} {g:10.1f}")
