"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd sqlite file: for every dispatch the counter's instances
(one row per XCD / shader engine) are summed, then dispatches of one kernel are averaged.  Prints a table and returns a
dict {kernel: {counter: value, "launches": n, "avg_ns": t, "instances": k}}."""
import collections, json, sqlite3, sys


def table(db_path, skip_first=0):
    """Kernels are keyed `name@<workgroups>`: one instantiation can serve launches of very different shapes (the
    64x128 GEMM runs the 512-block hidden layers and the 72-block output-layer backward of the policy)."""
    c = sqlite3.connect(db_path).cursor()
    blocks = {d: gx // max(1, wx) for d, gx, wx in c.execute("select dispatch_id, grid_x, workgroup_x from kernels")}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    inst = collections.defaultdict(lambda: collections.defaultdict(int))
    meta = {}
    for name, disp, ctr, val, dur in c.execute("select name, dispatch_id, counter_name, counter_value, duration from pmc_events"):
        per[disp][ctr] += val
        inst[disp][ctr] += 1
        meta[disp] = (name, dur)
    out = collections.OrderedDict()
    groups = collections.defaultdict(list)
    for disp in sorted(per):
        groups[(meta[disp][0], blocks.get(disp, 0))].append(disp)
    for (name, nblk), ds in groups.items():
        if "at::native" in name or "rocclr" in name:
            continue
        ds = ds[skip_first:] or ds
        k = name.split("(")[0].replace("void porl::", "").replace("porl::", "").replace(" ", "") + "@%d" % nblk
        e = {"launches": len(ds), "avg_ns": sum(meta[d][1] for d in ds) / len(ds)}
        for ctr in per[ds[0]]:
            e[ctr] = sum(per[d][ctr] for d in ds) / len(ds)
            e[ctr + "_instances"] = inst[ds[0]][ctr]
        out[k] = e
    return out


if __name__ == "__main__":
    t = table(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(json.dumps(t, indent=1))
