#!/bin/bash
# Run ON THE GPU BOX: the dispatches of the kernels whose name contains $1, last quarter of the run, one line each:
# grid size (threads), duration in us -- for finding which step of a schedule the time goes to.
export TMPDIR=/tmp
pat=$1; shift
R=$PWD
rm -rf $R/gpurun_out/kd
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kd -o t -- python3 $R/"$@" > $R/gpurun_out/kd.log 2>&1
cd $R
python3 - "$pat" <<PY
import csv, glob, sys
pat = sys.argv[1]
f = sorted(glob.glob("gpurun_out/kd/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
q = rows[len(rows) * 3 // 4:]
for r in q:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%-28s grid %9s  %8.1f us" % (r["Kernel_Name"].split("(")[0][-28:], r.get("Grid_Size", r.get("Grid_Size_X", "?")), d))
PY
rm -rf $R/gpurun_out/kd
