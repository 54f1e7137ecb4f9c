#!/bin/bash
# Rebuild the per-shape tuning cache from nothing (candidates timed with the caches flushed, ops.TUNE_FLUSH) for every bench
# workload, then compare with the cache it replaces.   bash tools/r03_retune.sh   (GPU box, via gpurun)
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03_retune"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
old="$root/heatnet-pub_amd/tune_cache.json"; new="$out/tune_cache.json"; rm -f "$new"
run() { python3 bench.py --no-cpu-baseline --no-alt "$@" 2>> "$out/err.txt"; }
echo "== tuning $(date +%T)"
export HEATNET_TUNE_CACHE="$new"
run --steps 2 --warmup 1 > /dev/null
run --steps 2 --warmup 1 --no-other-workloads --height 640 --width 1408 > /dev/null
run --steps 2 --warmup 1 --no-other-workloads --workload seg_bf16 --height 640 --width 1408 > /dev/null
run --steps 2 --warmup 1 --no-other-workloads --workload seg_resnext > /dev/null
echo "== compare $(date +%T)"
for wl in seg seg_bf16 confmax_seg confmax_critic; do for rep in 1 2; do
  for which in old new; do
    if [ $which = old ]; then export HEATNET_TUNE_CACHE="$out/old_copy.json"; cp "$old" "$out/old_copy.json"; else export HEATNET_TUNE_CACHE="$new"; fi
    run --steps 20 --warmup 4 --no-other-workloads --no-roofline --workload $wl > "$out/bench_${wl}_${which}_$rep.json"
    echo "$wl $which rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_${wl}_${which}_$rep.json" | head -1)"
  done
done; done
echo "== done $(date +%T)"
