#!/bin/bash
# Round-2 profile collection (run on the GPU box through gpurun, from the repo root):  bash tools/collect_profiles.sh <tag>
# Writes everything under gpurun_out/<tag>/; the summaries that are judged are then copied into profiles/ by hand.
tag=${1:-prof_r2}
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
step() { echo "== $1 $(date +%T)"; }

step "bench seg (headline, no profiler)"
python3 bench.py --steps 20 --warmup 5 > "$out/bench_seg.json" 2> "$out/bench_seg.err" || exit 1
step "bench seg_bf16"
python3 bench.py --steps 20 --warmup 5 --workload seg_bf16 --no-cpu-baseline > "$out/bench_seg_bf16.json" 2>> "$out/bench_seg.err" || exit 1
step "bench seg_bf16 640x1408"
python3 bench.py --steps 10 --warmup 3 --workload seg_bf16 --height 640 --width 1408 --no-cpu-baseline > "$out/bench_seg_bf16_640x1408.json" 2>> "$out/bench_seg.err" || exit 1
step "bench seg 640x1408"
python3 bench.py --steps 10 --warmup 3 --height 640 --width 1408 --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_seg_640x1408.json" 2>> "$out/bench_seg.err" || exit 1
step "bench confmax"
python3 bench.py --steps 10 --warmup 3 --workload confmax_seg --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_confmax_seg.json" 2>> "$out/bench_seg.err" || exit 1
python3 bench.py --steps 10 --warmup 3 --workload confmax_critic --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_confmax_critic.json" 2>> "$out/bench_seg.err" || exit 1

step "kernel stats seg"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_seg" -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads > "$out/stats_seg.json" 2> "$out/stats_seg.err" || exit 1
step "kernel stats seg_bf16"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_seg_bf16" -- python3 bench.py --steps 10 --warmup 4 --workload seg_bf16 --no-cpu-baseline > "$out/stats_seg_bf16.json" 2> "$out/stats_seg_bf16.err" || exit 1

for wl in seg seg_bf16; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    step "pmc $ctr $wl"
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/pmc_${wl}_$ctr" -- python3 bench.py --steps 2 --warmup 1 --workload $wl --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/pmc_${wl}_$ctr.log" 2>&1 || exit 1
  done
done
step "done"
