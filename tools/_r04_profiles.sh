# round-4 evidence: run on the GPU box; summaries land under gpurun_out/, copied into profiles/ afterwards
set -e
timeout -k 10 900 bash tools/profile_bench.sh r04_bench bench.py --skip-cpu --skip-configs --mode tiled > gpurun_out/r04_bench.log 2>&1 || { tail -20 gpurun_out/r04_bench.log; exit 1; }
grep -h "^{" gpurun_out/r04_bench/trace.stdout | tail -1 > gpurun_out/r04_bench/r04_bench_traced_run.json
timeout -k 10 300 bash tools/profile_cmd.sh r04_factor tools/time_factor_abi.py 78125 64 5 1 > gpurun_out/r04_factor.log 2>&1 || { tail -20 gpurun_out/r04_factor.log; exit 1; }
cp gpurun_out/r04_factor/trace.stdout gpurun_out/r04_factor/r04_factor_traced_run.txt
timeout -k 10 300 bash tools/trace_cmd.sh r04_spgemm bench_configs.py --only spgemm --skip-cpu > gpurun_out/r04_spgemm.log 2>&1 || { tail -20 gpurun_out/r04_spgemm.log; exit 1; }
timeout -k 10 300 bash tools/trace_cmd.sh r04_lusolve bench_configs.py --only lusolve --skip-cpu > gpurun_out/r04_lusolve.log 2>&1 || { tail -20 gpurun_out/r04_lusolve.log; exit 1; }
grep -h '^{"config' gpurun_out/r04_spgemm/trace.stdout gpurun_out/r04_lusolve/trace.stdout > gpurun_out/r04_configs_traced_runs.jsonl
timeout -k 10 60 tools/ubench/valu_rate > gpurun_out/r04_valu_rate.txt 2>&1
timeout -k 10 60 tools/ubench/scatter_mall > gpurun_out/r04_scatter_mall.txt 2>&1
timeout -k 10 120 python tools/time_lu_bordered.py 40 400 4000 > gpurun_out/r04_lu_bordered.txt 2>&1
ls gpurun_out/r04_bench gpurun_out/r04_factor gpurun_out/r04_spgemm gpurun_out/r04_lusolve
