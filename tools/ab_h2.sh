set -e
for r in 1 2; do
for s in 0 0,3; do
  HEATNET_H2_SCHEDULES=$s HEATNET_TUNE_CACHE=/tmp/tc_h2_$s.json timeout -k 10 280 python bench.py --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$s', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['conv_ms_per_step'])"
done; done
