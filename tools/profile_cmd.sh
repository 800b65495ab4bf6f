#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + HBM counters for ANY python command of this repo (the
# bench_configs.py sections, the tools/time_*.py scripts), reduced to <tag>_kernel_stats.csv and <tag>_traffic.json under
# gpurun_out/<tag>/ (copy them into profiles/ and commit).  Like tools/profile_bench.sh, without bench.py's --steps flag.
#   tools/profile_cmd.sh r03_spgemm bench_configs.py --only spgemm --skip-cpu
# Passes: (1) --kernel-trace --stats, (2) --pmc FETCH_SIZE, (3) --pmc WRITE_SIZE -- counters in their own runs, never
# together with a trace (MI355X_MICROARCH.md).  The program itself follows `--`.
set -e
tag=$1; shift
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/"$@" > $out/trace.stdout 2> $out/trace.stderr
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 $root/"$@" > $out/fetch.stdout 2> $out/fetch.stderr
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 $root/"$@" > $out/write.stdout 2> $out/write.stderr
cd $root
python3 tools/profile_reduce.py $out $tag
rm -rf $out/trace $out/fetch $out/write
ls -la $out
