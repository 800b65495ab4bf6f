#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the instruction-issue counters of the kernels DESIGN.md calls issue-bound -- VALUBusy,
# SQ_INSTS_VALU, SQ_VALU_MFMA_BUSY_CYCLES, SQ_ACTIVE_INST_LDS, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE -- in passes of their own (never with
# a trace), reduced by tools/profile_issue_reduce.py to gpurun_out/<tag>/<tag>_issue.json.
#   tools/profile_issue.sh r05_issue_exact tools/time_mfma_solve.py 78125 64 128 3 3 1
set -e
tag=$1; shift
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --pmc VALUBusy SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -o a -- python3 $root/"$@" > $out/p1.stdout 2> $out/p1.stderr
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $out/p2 -o b -- python3 $root/"$@" > $out/p2.stdout 2> $out/p2.stderr
cd $root
python3 tools/profile_issue_reduce.py $out $tag
rm -rf $out/p1 $out/p2
ls -la $out
