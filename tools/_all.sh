for mf in 448 384 320; do
for w in sphere2500 manhattan3500 synthetic10k; do
  SPP_MID_FRONT_MAX=$mf python bench.py --workload $w --no-cpu-baseline --steps 30 > gpurun_out/bench_tmp.log 2>&1
  grep '^{' gpurun_out/bench_tmp.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$mf $w', round(d['ms_per_step'],3), d['phase_ms']['factor'], d['phase_ms']['trisolve'])"
done; done
