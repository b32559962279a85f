set -e
bash tools/profile_round.sh r03
for w in ladybug49 sphere2500 manhattan3500 synthetic10k; do
  python bench.py --workload $w > gpurun_out/bench_$w.log 2>&1
done
python bench.py > gpurun_out/bench_default.log 2>&1
grep '^{' gpurun_out/bench_default.log | tail -1 | cut -c1-330
