#!/bin/bash
# Regenerates the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01
# kernel trace + stats of the default bench, then two separate PMC passes (FETCH_SIZE / WRITE_SIZE can not
# share a pass), all written under gpurun_out/; tools/pmc_summary.py folds them into profiles/.
set -e
tag=${1:-r01}
out=$PWD/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o run --output-format csv -- python3 bench.py --steps 20 --warmup 3 > $out/${tag}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_pmc_write -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_write.log 2>&1
echo done
