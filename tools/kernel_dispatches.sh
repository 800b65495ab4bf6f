#!/bin/bash
# Run ON THE GPU BOX: every dispatch of the kernels whose name contains $1 (rocprofv3 --kernel-trace), as
# "grid workgroup duration_us", sorted by start time, for one python command.
#   tools/kernel_dispatches.sh k_chol_coop tools/time_chol_nd.py 700
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
half = rows[len(rows) // 2:]          # the second (timed) factorisation
print("dispatches", len(rows), "second half", len(half))
tot = {}
for r in half:
    g = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = "1 wg" if g == 1 else ("2-8 wg" if g <= 8 else ("9-63 wg" if g < 64 else "64+ wg"))
    t = tot.setdefault(key, [0, 0.0, 0.0])
    t[0] += 1; t[1] += d; t[2] = max(t[2], d)
for k, (c, t, m) in sorted(tot.items()):
    print("%-8s launches %5d  total %9.1f us  avg %8.1f  max %8.1f" % (k, c, t, t / c, m))
PY
rm -rf gpurun_out/kd
