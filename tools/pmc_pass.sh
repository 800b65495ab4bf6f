#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass (counters given as $1, space separated) over a python command; prints per
# kernel name the summed counter values.  Counters in their own run, never together with a trace.
#   tools/pmc_pass.sh "SQ_INSTS_VALU SQ_INSTS_SALU" tools/time_spgemm.py
export TMPDIR=/tmp
R=$PWD
ctr=$1; shift
rm -rf $R/gpurun_out/pmc
cd /tmp && timeout -k 10 400 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc -o p -- python3 $R/"$@" > $R/gpurun_out/pmc.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/pmc/**/*counter_collection.csv", recursive=True))
if not f:
    print(open("gpurun_out/pmc.log").read()[-2000:])
    raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f[-1])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:6]:
    print(k)
    for c, v in d.items():
        print("   %-28s %.4g  (%d dispatches)" % (c, v, n[(k, c)]))
PY
rm -rf gpurun_out/pmc
