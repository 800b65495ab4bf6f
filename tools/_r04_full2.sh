set -e
mkdir -p gpurun_out/r04g
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04g/gputests.log 2>&1 || { tail -40 gpurun_out/r04g/gputests.log; exit 1; }
tail -3 gpurun_out/r04g/gputests.log
timeout -k 10 400 python tools/fuzz_campaign.py 240 4100000 > gpurun_out/r04g/fuzz.out 2>&1 || { tail -30 gpurun_out/r04g/fuzz.out; exit 1; }
tail -2 gpurun_out/r04g/fuzz.out
timeout -k 10 600 python bench.py > gpurun_out/r04g/bench.json 2> gpurun_out/r04g/bench.err || { tail -20 gpurun_out/r04g/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04g/bench.json') if l.startswith('{')][-1])
print(json.dumps({k:d[k] for k in ('metric','value','ms_per_step','roofline')}))
c=d['cholsol']
print(json.dumps({k:c[k] for k in ('solves_per_s','ms_per_batch','factor_s','chol_roofline','end_to_end_solves_per_s_per_gpu','exact_order')}, indent=1))
PY
