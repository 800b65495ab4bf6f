# round-5 evidence at the final code: run on the GPU box; summaries land under gpurun_out/, copied into profiles/ afterwards
set -e
timeout -k 10 900 bash tools/profile_bench.sh r05_bench bench.py --skip-cpu --skip-configs --mode tiled > gpurun_out/r05_bench.log 2>&1 || { tail -20 gpurun_out/r05_bench.log; exit 1; }
grep -h "^{" gpurun_out/r05_bench/trace.stdout | tail -1 > gpurun_out/r05_bench/r05_bench_traced_run.json
timeout -k 10 300 bash tools/profile_cmd.sh r05_factor tools/time_factor_fused.py 78125 64 5 1 > gpurun_out/r05_factor.log 2>&1 || { tail -20 gpurun_out/r05_factor.log; exit 1; }
cp gpurun_out/r05_factor/trace.stdout gpurun_out/r05_factor/r05_factor_traced_run.txt
timeout -k 10 300 bash tools/profile_cmd.sh r05_factor_mfma tools/time_factor_fused.py 78125 64 5 0 > gpurun_out/r05_factor_mfma.log 2>&1 || { tail -20 gpurun_out/r05_factor_mfma.log; exit 1; }
cp gpurun_out/r05_factor_mfma/trace.stdout gpurun_out/r05_factor_mfma/r05_factor_mfma_traced_run.txt
timeout -k 10 400 bash tools/profile_cmd.sh r05_ragged tools/time_ragged_cliques.py > gpurun_out/r05_ragged.log 2>&1 || { tail -20 gpurun_out/r05_ragged.log; exit 1; }
cp gpurun_out/r05_ragged/trace.stdout gpurun_out/r05_ragged/r05_ragged_traced_run.txt
timeout -k 10 400 bash tools/profile_cmd.sh r05_lusolve bench_configs.py --only lusolve --skip-cpu > gpurun_out/r05_lusolve.log 2>&1 || { tail -20 gpurun_out/r05_lusolve.log; exit 1; }
grep -h '^{"config' gpurun_out/r05_lusolve/trace.stdout > gpurun_out/r05_lusolve/r05_lusolve_traced_run.jsonl
timeout -k 10 400 bash tools/profile_cmd.sh r05_spgemm bench_configs.py --only spgemm --skip-cpu > gpurun_out/r05_spgemm.log 2>&1 || { tail -20 gpurun_out/r05_spgemm.log; exit 1; }
grep -h '^{"config' gpurun_out/r05_spgemm/trace.stdout > gpurun_out/r05_spgemm/r05_spgemm_traced_run.jsonl
timeout -k 10 400 bash tools/profile_cmd.sh r05_transpose bench_configs.py --only transpose --skip-cpu > gpurun_out/r05_transpose.log 2>&1 || { tail -20 gpurun_out/r05_transpose.log; exit 1; }
grep -h '^{"config' gpurun_out/r05_transpose/trace.stdout > gpurun_out/r05_transpose/r05_transpose_traced_run.jsonl
ls gpurun_out/r05_bench gpurun_out/r05_factor gpurun_out/r05_factor_mfma gpurun_out/r05_ragged gpurun_out/r05_lusolve gpurun_out/r05_spgemm gpurun_out/r05_transpose
