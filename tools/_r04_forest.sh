#!/bin/bash
# round 4: forests of small sparse trees -- tests of the block paths, then the timing tool
set -o pipefail
mkdir -p gpurun_out/r04f
timeout -k 10 600 python -m pytest tests/test_gpu_cholclique.py tests/test_gpu_cholesky.py tests/test_gpu_orders.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r04f/tests.log 2>&1 || { tail -40 gpurun_out/r04f/tests.log; exit 1; }
tail -3 gpurun_out/r04f/tests.log
CSX_CHOL_TIMING=1 timeout -k 10 300 python tools/time_forest_sparse.py 200000 24 3 > gpurun_out/r04f/forest.log 2>&1 || { tail -20 gpurun_out/r04f/forest.log; exit 1; }
cat gpurun_out/r04f/forest.log
