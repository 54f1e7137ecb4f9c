#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03h"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== tests $(date +%T)"
timeout -k 10 300 python3 -m pytest tests/test_gpu_resnext.py tests/test_gpu_planes.py -q -k "resnext or custom or fused" > "$out/tests.log" 2>&1; tail -4 "$out/tests.log"
echo "== main-high A/B $(date +%T)"
for rep in 1 2 3; do for mh in 0 1; do
  HEATNET_MAIN_HIGH=$mh timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_mh${mh}_$rep.json" 2>> "$out/bench.err"
  echo "main_high=$mh rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_mh${mh}_$rep.json" | head -1)"
done; done
echo "== kernel stats $(date +%T)"
for wl in seg_bf16 confmax_critic confmax_seg; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$wl" -- python3 bench.py --steps 6 --warmup 2 --workload $wl --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/stats_$wl.json" 2> "$out/stats_$wl.err"
  echo "$wl: $(grep -o '"ms_per_step": [0-9.]*' "$out/stats_$wl.json" | head -1)"
done
echo "== done $(date +%T)"
