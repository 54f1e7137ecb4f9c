#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03j"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== bf16 tests $(date +%T)"
timeout -k 10 400 python3 -m pytest tests/test_gpu_planes.py tests/test_gpu_network.py -q -k "bf16" > "$out/tests_bf16.log" 2>&1; rc=$?
grep -E "^FAILED|passed|failed" "$out/tests_bf16.log" | tail -30
[ $rc -gt 1 ] && exit 1
echo "== bf16 y A/B $(date +%T)"
for rep in 1 2; do for yb in 0 1; do
  HEATNET_BF16_Y=$yb timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --workload seg_bf16 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_bf16_y${yb}_$rep.json" 2>> "$out/bench.err" || exit 1
  echo "bf16_y=$yb rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_bf16_y${yb}_$rep.json" | head -1)"
done; done
echo "== kernel stats $(date +%T)"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_seg_bf16" -- python3 bench.py --steps 6 --warmup 2 --workload seg_bf16 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/stats_seg_bf16.json" 2> "$out/stats_seg_bf16.err"
echo "== done $(date +%T)"
