#!/bin/bash
# Timing experiments on k_gaxpy_tiled (needs `python csparse.py_amd/build.py --ablation`): one bench.py run per
# code, HIP-event ms per pass of the headline matrix.  code = VARIANT + 100 * waves + 10000 * groups-per-step.
# usage: tools/ablate_tiled.sh "0 1200 31200" [extra bench args]
export CSX_LIB=$PWD/csparse.py_amd/libcsx_ablation.so
for v in $1; do
  CSX_TILED_VARIANT=$v python bench.py --mode tiled --skip-gspd --skip-cpu --steps 20 --warmup 3 ${@:2} 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('code %6s: %.4f ms/pass  frac %.3f' % ('$v', d['roofline']['step_ms_hip_events'], d['roofline']['frac']))"
done
