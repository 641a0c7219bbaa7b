"""Print the launch sequence of ONE fp32 encoder forward from a rocprofv3 --kernel-trace CSV (name, grid, us)."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last forward = from the last patch_stats launch... take the last two patch_bn launches
idx = [i for i, n in enumerate(names) if "patch_stats" in n]
lo = idx[-2]
for r in rows[lo:]:
    n = r["Kernel_Name"]
    n = n.replace("void porl::", "").replace("porl::", "")[:60]
    print(f"{n:62s} grid {r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size')} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size'))} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us")
