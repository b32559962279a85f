#!/bin/bash
# sweep of the S accumulation's items-per-wave (SPP_SACC_CHUNK) and item tile (SPP_SACC_TILE) on the Venice shape:
# prints ms_per_step and the phase times of bench.py for each setting
mkdir -p gpurun_out
for u in ${ULMS:-1}; do export SPP_SACC_ULM=$u; for tc in ${TCOLS:--2}; do export SPP_SACC_TILE_COLS=$tc; for x in ${XCDS:-1}; do export SPP_SACC_XCD=$x; for c in ${CHUNKS:-8}; do for t in ${TILES:-4}; do
  SPP_SACC_CHUNK=$c SPP_SACC_TILE=$t python bench.py --steps 10 --warmup 2 --no-cpu-baseline --cpu-cholmod off > gpurun_out/sw.json 2>/dev/null || exit 1
  python - "$c" "$t" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1])
p=d['phase_ms']
print("ulm",__import__("os").environ.get("SPP_SACC_ULM"),"tcols",__import__("os").environ.get("SPP_SACC_TILE_COLS"),"xcd",__import__("os").environ.get("SPP_SACC_XCD"),"chunk",sys.argv[1],'tile',sys.argv[2],'step %.3f'%d['ms_per_step'],'inv %.3f gemm %.3f factor %.3f tri %.3f back %.3f'%(p['schur_inv'],p['schur_gemm'],p['factor'],p['trisolve'],p['backsubst']), 'norm',d['solution_norm'], flush=True)
PY
done; done; done; done; done
