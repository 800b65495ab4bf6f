#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 --kernel-trace --stats of one python command -> gpurun_out/<tag>/<tag>_kernel_stats.csv
# (no counter passes: with thousands of short launches the PMC passes take minutes and rocprofv3 has crashed on them).
#   tools/trace_cmd.sh r03_connected bench_configs.py --only connected --skip-cpu
set -e
tag=$1; shift
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/"$@" > $out/trace.stdout 2> $out/trace.stderr
cd $root
python3 tools/profile_reduce.py $out $tag
rm -rf $out/trace $out/${tag}_traffic.json
ls $out
