R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; mkdir -p $O
cd $R
summ() { python - "$1" <<'P'
import re,sys
t=[float(m.group(1)) for m in re.finditer(r"\(\+\s*([\d.]+)\)", open(sys.argv[1]).read())]
tail=[l for l in open(sys.argv[1]) if l.startswith("updates")]
av=lambda a,b: sum(t[a:b])/max(1,len(t[a:b]))
print(sys.argv[1].split('/')[-1], "upd1-5 %.0f  6-10 %.0f  11-15 %.0f  16-20 %.0f  21-25 %.0f  26-30 %.0f |"%(av(1,6),av(6,11),av(11,16),av(16,21),av(21,26),av(26,30)), " ".join(x.split(';')[0].replace('updates ','') for x in tail))
P
}
python -c "import torch; print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
python scripts/bench_ramp.py > $O/p_base.log 2>&1; summ $O/p_base.log
MAIN_PRIORITY=-1 python scripts/bench_ramp.py > $O/p_main.log 2>&1; summ $O/p_main.log
MAIN_PRIORITY=0 python scripts/bench_ramp.py > $O/p_main0.log 2>&1; summ $O/p_main0.log
PORL_SIDE_PRIORITY=1 python scripts/bench_ramp.py > $O/p_sidelow.log 2>&1; summ $O/p_sidelow.log
python scripts/bench_ramp.py > $O/p_base2.log 2>&1; summ $O/p_base2.log
