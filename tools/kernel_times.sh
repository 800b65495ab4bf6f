#!/bin/bash
# Run ON THE GPU BOX: per-kernel times (rocprofv3 --kernel-trace --stats) of one python command, printed as a table.
#   tools/kernel_times.sh bench_configs.py --only transpose --skip-cpu
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/kt
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt -o t -- python3 $R/"$@" > $R/gpurun_out/kt.log 2>&1
cd $R
find gpurun_out/kt -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/kt/**/*kernel_stats.csv", recursive=True))[-1]
print("%-78s %5s %10s %10s %10s" % ("kernel", "calls", "avg us", "min us", "max us"))
for r in csv.DictReader(open(f)):
    print("%-78s %5s %10.1f %10.1f %10.1f" % (r["Name"][:78], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
tail -2 gpurun_out/kt.log | cut -c1-400
