#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04f
CSX_CHOL_TIMING=1 timeout -k 10 300 python tools/time_forest_sparse.py 200000 24 3 > gpurun_out/r04f/forest_laps.log 2>&1 || { tail -20 gpurun_out/r04f/forest_laps.log; exit 1; }
cat gpurun_out/r04f/forest_laps.log
