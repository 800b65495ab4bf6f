# round 4, run on the GPU box through gpurun: the whole GPU suite, a fuzz campaign and bench.py at HEAD
set -e
mkdir -p gpurun_out/r04_full
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_full/gputests.log 2>&1 || { tail -40 gpurun_out/r04_full/gputests.log; exit 1; }
tail -3 gpurun_out/r04_full/gputests.log
timeout -k 10 400 python tools/fuzz_campaign.py 240 4100000 > gpurun_out/r04_full/fuzz.out 2>&1 || { tail -30 gpurun_out/r04_full/fuzz.out; exit 1; }
tail -2 gpurun_out/r04_full/fuzz.out
timeout -k 10 600 python bench.py > gpurun_out/r04_full/bench.json 2> gpurun_out/r04_full/bench.err || { tail -20 gpurun_out/r04_full/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_full/bench.json') if l.startswith('{')][-1])
print(json.dumps({k:d[k] for k in ('metric','value','ms_per_step','roofline')}))
c=d['cholsol']
print(json.dumps({k:c[k] for k in ('solves_per_s','ms_per_batch','factor_s','chol_roofline','end_to_end_solves_per_s_per_gpu','exact_order')}, indent=1))
PY
