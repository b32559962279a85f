#!/bin/bash
# Regenerates the rocprofv3 evidence of a round for a list of workloads (run through gpurun from the repo root):
#   tools/profile_all.sh r02 venice871 ladybug49 sphere2500 manhattan3500
# per workload: the plain bench line, rocprofv3 --kernel-trace --stats of the same command (per-kernel durations), and
# two separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass; --kernel-trace only beside --pmc), folded by
# tools/pmc_summary.py. Everything is written under gpurun_out/<tag>_*; the summaries to commit go to profiles/<tag>_*.
set -e
tag=$1; shift
out=$PWD/gpurun_out
mkdir -p $out profiles
export TMPDIR=/tmp
for w in "$@"; do
	case $w in
		venice871) dom=dense_tail_kernel; steps=20;;
		ladybug49) dom=s_accum_kernel; steps=50;;
		synthetic10k) dom=s_accum_kernel; steps=5;;
		*) dom=front_dag_kernel; steps=50;;
	esac
	echo "== $w"
	python3 bench.py --workload $w --steps $steps --warmup 3 > profiles/${tag}_${w}_bench_line.json 2> $out/${tag}_${w}_bench.err
	rocprofv3 --kernel-trace --stats -d $out/${tag}_${w}_trace -o run --output-format csv -- python3 bench.py --workload $w --steps $steps --warmup 3 --no-cpu-baseline > profiles/${tag}_${w}_bench_line_under_rocprof.json 2> $out/${tag}_${w}_trace.err
	cp $out/${tag}_${w}_trace/run_kernel_stats.csv profiles/${tag}_${w}_kernel_stats.csv
	rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_${w}_pmc_fetch -o run --output-format csv -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_${w}_pmc_fetch.log 2>&1
	rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_${w}_pmc_write -o run --output-format csv -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_${w}_pmc_write.log 2>&1
	python3 tools/pmc_summary.py $out/${tag}_${w}_pmc_fetch $out/${tag}_${w}_pmc_write $w $dom profiles/pmc_traffic.json profiles/${tag}_${w}_pmc_fetch_write.csv $tag > $out/${tag}_${w}_pmc_summary.log
done
echo done
