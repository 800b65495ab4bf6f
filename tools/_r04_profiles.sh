# round-4 evidence at the final code: run on the GPU box; summaries land under gpurun_out/, copied into profiles/ afterwards
set -e
timeout -k 10 900 bash tools/profile_bench.sh r04_bench bench.py --skip-cpu --skip-configs --mode tiled > gpurun_out/r04_bench.log 2>&1 || { tail -20 gpurun_out/r04_bench.log; exit 1; }
grep -h "^{" gpurun_out/r04_bench/trace.stdout | tail -1 > gpurun_out/r04_bench/r04_bench_traced_run.json
timeout -k 10 300 bash tools/profile_cmd.sh r04_factor tools/time_factor_abi.py 78125 64 5 1 > gpurun_out/r04_factor.log 2>&1 || { tail -20 gpurun_out/r04_factor.log; exit 1; }
cp gpurun_out/r04_factor/trace.stdout gpurun_out/r04_factor/r04_factor_traced_run.txt
timeout -k 10 300 bash tools/profile_cmd.sh r04_forest tools/time_forest_sparse.py 200000 24 3 > gpurun_out/r04_forest.log 2>&1 || { tail -20 gpurun_out/r04_forest.log; exit 1; }
cp gpurun_out/r04_forest/trace.stdout gpurun_out/r04_forest/r04_forest_traced_run.txt
timeout -k 10 300 bash tools/trace_cmd.sh r04_lusolve bench_configs.py --only lusolve --skip-cpu > gpurun_out/r04_lusolve.log 2>&1 || { tail -20 gpurun_out/r04_lusolve.log; exit 1; }
grep -h '^{"config' gpurun_out/r04_lusolve/trace.stdout > gpurun_out/r04_lusolve_traced_run.jsonl
ls gpurun_out/r04_bench gpurun_out/r04_factor gpurun_out/r04_forest gpurun_out/r04_lusolve
