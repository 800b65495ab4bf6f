set -e
mkdir -p gpurun_out/r04c
timeout -k 10 900 python -m pytest tests/test_gpu_cholesky.py tests/test_lu_oracle.py tests/test_gpu_qrsol.py tests/test_gpu_multiply.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r04c/tests.log 2>&1 || { tail -60 gpurun_out/r04c/tests.log; exit 1; }
tail -3 gpurun_out/r04c/tests.log
timeout -k 10 400 python tools/fuzz_campaign.py 240 3100000 > gpurun_out/r04c/fuzz.out 2>&1 || { tail -30 gpurun_out/r04c/fuzz.out; exit 1; }
tail -2 gpurun_out/r04c/fuzz.out
