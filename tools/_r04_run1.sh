set -e
mkdir -p gpurun_out/r04a
timeout -k 10 900 python -m pytest tests/test_gpu_cholclique.py tests/test_gpu_cholesky.py -x -q > gpurun_out/r04a/tests.log 2>&1 || { tail -40 gpurun_out/r04a/tests.log; exit 1; }
tail -3 gpurun_out/r04a/tests.log
CSX_CHOL_TIMING=1 timeout -k 10 300 python tools/time_factor_abi.py 78125 64 3 1 > gpurun_out/r04a/factor_clique.log 2>&1
cat gpurun_out/r04a/factor_clique.log
CSX_CHOL_TIMING=1 timeout -k 10 300 python tools/time_factor_abi.py 78125 64 2 0 > gpurun_out/r04a/factor_general.log 2>&1
tail -30 gpurun_out/r04a/factor_general.log
