set -e
mkdir -p gpurun_out/r04s
timeout -k 10 300 python tools/time_w.py > gpurun_out/r04s/w.log 2>&1 || { tail -20 gpurun_out/r04s/w.log; exit 1; }
cat gpurun_out/r04s/w.log
timeout -k 10 300 python tools/time_forest_sparse.py 200000 24 2 > gpurun_out/r04s/forest.log 2>&1 || { tail -20 gpurun_out/r04s/forest.log; exit 1; }
tail -2 gpurun_out/r04s/forest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04s/prof -o w -- python3 $GRAFT_REPO_ROOT/tools/time_w.py > $GRAFT_REPO_ROOT/gpurun_out/r04s/prof.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/r04s/prof.log; exit 1; }
cd $GRAFT_REPO_ROOT
find gpurun_out/r04s/prof -name "*kernel_stats*" | head -3
python - <<'PY'
import glob, csv
for f in glob.glob('gpurun_out/r04s/prof/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(r['Name'][:90], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
