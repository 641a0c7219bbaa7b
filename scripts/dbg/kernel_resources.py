"""VGPR / LDS / duration per distinct kernel from a rocprofv3 --kernel-trace database (rocpd)."""
import sqlite3, sys, glob
db = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]
c = sqlite3.connect(db)
seen = {}
for name, vg, ag, lds, slds, dur in c.execute("select name, vgpr_count, accum_vgpr_count, lds_size, static_lds_size, duration from kernels"):
    k = name.replace("void porl::", "").replace("porl::", "")[:70]
    e = seen.setdefault(k, [vg, ag, lds, slds, 0, 0.0])
    e[4] += 1; e[5] += dur / 1e3
for k, (vg, ag, lds, slds, n, us) in sorted(seen.items(), key=lambda kv: -kv[1][5])[:16]:
    print(f"{k:72s} vgpr {vg:4d} agpr {ag:3d} lds {lds:6d} n {n:5d} avg {us / n:9.1f} us")
