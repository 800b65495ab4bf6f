#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + HBM counters for a command, reduced to small
# CSV/JSON summaries under gpurun_out/<tag>/ that are then copied into profiles/ and committed.
#   tools/profile_bench.sh <tag> <python script + args ...>       e.g.  tools/profile_bench.sh r02_bench bench.py --skip-cpu --skip-configs --mode tiled
# (--mode tiled: without it bench.py also TRIES the wave kernel on G-rand, and the per-kernel counter summary of
# k_gaxpy_rows4 would mix those launches with the G-spd ones it is quoted for)
# Passes: (1) --kernel-trace --stats, (2) --pmc FETCH_SIZE, (3) --pmc WRITE_SIZE -- counters in their own runs,
# never together with a trace (MI355X_MICROARCH.md, rocprofv3 PMC slots).  The program itself follows `--`.
set -e
tag=$1; shift
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/"$@" > $out/trace.stdout 2> $out/trace.stderr
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 $root/"$@" --steps 5 > $out/fetch.stdout 2> $out/fetch.stderr
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 $root/"$@" --steps 5 > $out/write.stdout 2> $out/write.stderr
cd $root
python3 tools/profile_reduce.py $out $tag
# keep only the summaries (raw traces are large)
rm -rf $out/trace $out/fetch $out/write
ls -la $out
