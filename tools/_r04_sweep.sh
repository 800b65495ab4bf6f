# round 4: the fused-sweep tests and the two timing tools (W, the sparse forest)
set -e
mkdir -p gpurun_out/r04s
timeout -k 10 600 python -m pytest tests/test_gpu_trisolve.py tests/test_gpu_cholesky.py tests/test_gpu_configs.py tests/test_gpu_cholclique.py tests/test_gpu_comm.py -x -q -m gpu > gpurun_out/r04s/tests.log 2>&1 || { tail -40 gpurun_out/r04s/tests.log; exit 1; }
tail -2 gpurun_out/r04s/tests.log
timeout -k 10 300 python tools/time_w.py > gpurun_out/r04s/w.log 2>&1 || { tail -20 gpurun_out/r04s/w.log; exit 1; }
cat gpurun_out/r04s/w.log
timeout -k 10 300 python tools/time_forest_sparse.py 200000 24 2 > gpurun_out/r04s/forest.log 2>&1 || { tail -20 gpurun_out/r04s/forest.log; exit 1; }
tail -2 gpurun_out/r04s/forest.log
