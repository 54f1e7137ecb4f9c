#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03k"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== conv tables $(date +%T)"
for wl in confmax_critic confmax_seg; do
  HN_CONV_TABLE="$out/conv_table_$wl.txt" timeout -k 10 240 python3 bench.py --steps 6 --warmup 2 --workload $wl --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_$wl.json" 2>> "$out/bench.err" || exit 1
  echo "$wl: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_$wl.json" | head -1)"
done
echo "== full suite $(date +%T)"
timeout -k 10 800 python3 -m pytest tests -m gpu -q --durations=60 > "$out/tests_full.log" 2>&1; rc=$?
grep -E "^FAILED|passed|failed" "$out/tests_full.log" | tail -10
echo "== done rc=$rc $(date +%T)"
