#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03o"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== host enqueue time $(date +%T)"
timeout -k 10 200 python3 tools/host_time.py seg 2>/dev/null | tail -1
timeout -k 10 200 python3 tools/host_time.py seg_bf16 2>/dev/null | tail -1
echo "== dist path A/B $(date +%T)"
for rep in 1 2; do
for cfg in "0 1" "1 1" "1 0"; do
  set -- $cfg
  HN_BENCH_FORCE_DIST=$1 HEATNET_WGRAD_STREAM=$2 timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_dist$1_side$2_$rep.json" 2>> "$out/bench.err" || exit 1
  echo "force_dist=$1 side=$2 rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_dist$1_side$2_$rep.json" | head -1)"
done; done
echo "== dist path kernel stats $(date +%T)"
HN_BENCH_FORCE_DIST=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_dist" -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/stats_dist.json" 2> "$out/stats_dist.err"
echo "== done $(date +%T)"
