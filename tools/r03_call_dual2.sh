#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03dual2"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== full suite (dual streams on) $(date +%T)"
timeout -k 10 900 python3 -m pytest tests -m gpu -q > "$out/tests_full.log" 2>&1; rc=$?
grep -E "^FAILED|passed|failed" "$out/tests_full.log" | tail -8
echo "== A/B $(date +%T)"
for rep in 1 2; do for du in 0 1; do
  for wl in seg seg_bf16 confmax_seg; do
    HEATNET_DUAL_STREAMS=$du timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --workload $wl --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_${wl}_dual${du}_$rep.json" 2>> "$out/bench.err" || exit 1
    echo "dual=$du $wl rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_${wl}_dual${du}_$rep.json" | head -1)"
  done
  HN_BENCH_FORCE_DIST=1 HEATNET_DUAL_STREAMS=$du timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_dist_dual${du}_$rep.json" 2>> "$out/bench.err" || exit 1
  echo "dual=$du N>1 path rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_dist_dual${du}_$rep.json" | head -1)"
done; done
echo "== done rc=$rc $(date +%T)"
exit $rc
