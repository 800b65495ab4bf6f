set -e
mkdir -p gpurun_out/r04d
timeout -k 10 900 python -m pytest tests/test_gpu_cholclique.py tests/test_gpu_cholesky.py tests/test_gpu_lu_etree.py tests/test_gpu_multirank.py tests/test_gpu_comm.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r04d/tests.log 2>&1 || { tail -60 gpurun_out/r04d/tests.log; exit 1; }
tail -3 gpurun_out/r04d/tests.log
timeout -k 10 300 python bench.py --dry-exchange --force-sharded > gpurun_out/r04d/dry.json 2> gpurun_out/r04d/dry.err || { tail -20 gpurun_out/r04d/dry.err; exit 1; }
cat gpurun_out/r04d/dry.json
CSX_SINGLE_DEVICE=1 CSX_COMM_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 3 --dry-exchange > gpurun_out/r04d/dry3.json 2> gpurun_out/r04d/dry3.err || { tail -20 gpurun_out/r04d/dry3.err; exit 1; }
cat gpurun_out/r04d/dry3.json
timeout -k 10 600 python bench_configs.py --only lusolve --skip-cpu > gpurun_out/r04d/lusolve.json 2> gpurun_out/r04d/lusolve.err || { tail -20 gpurun_out/r04d/lusolve.err; exit 1; }
cat gpurun_out/r04d/lusolve.json | cut -c1-3000
