#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03i"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== small-conv tests $(date +%T)"
timeout -k 10 240 python3 -m pytest tests/test_gpu_small_convs.py tests/test_gpu_kernels.py -q -x -k "small or stem or narrow or leaky" > "$out/tests_small.log" 2>&1 || { tail -30 "$out/tests_small.log"; exit 1; }
tail -2 "$out/tests_small.log"
echo "== stem A/B $(date +%T)"
for rep in 1 2; do for st in 0 1; do
  HEATNET_SMALL_STEM=$st timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_stem${st}_$rep.json" 2>> "$out/bench.err" || exit 1
  echo "stem=$st rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_stem${st}_$rep.json" | head -1)"
done; done
echo "== other workloads $(date +%T)"
for wl in confmax_critic seg_bf16; do
  timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --workload $wl --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_$wl.json" 2>> "$out/bench.err" || exit 1
  echo "$wl: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_$wl.json" | head -1)"
done
echo "== full suite $(date +%T)"
timeout -k 10 700 python3 -m pytest tests -m gpu -q -x > "$out/tests_full.log" 2>&1; rc=$?
tail -5 "$out/tests_full.log"
echo "== done rc=$rc $(date +%T)"
exit $rc
