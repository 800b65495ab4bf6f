# ablation build: what each phase of k_chol_clique costs (CSX_CQ_PARTS: 1 load, 2 factor, 4 store)
set -e
export CSX_LIB=$PWD/csparse.py_amd/libcsx_ablation.so
for parts in 1 5 7; do
  echo "== parts $parts"
  CSX_CQ_PARTS=$parts CSX_CHOL_TIMING=1 timeout -k 10 120 python tools/time_factor_abi.py 78125 64 3 1 2>&1 | grep "numeric" | tail -2
done
