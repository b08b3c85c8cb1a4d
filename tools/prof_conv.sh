cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/convprof -o conv -- python $GRAFT_REPO_ROOT/tools/microbench_conv.py > $GRAFT_REPO_ROOT/gpurun_out/convprof.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/convprof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}  pct {r['Percentage']}")
PY
tail -2 $GRAFT_REPO_ROOT/gpurun_out/convprof.log
