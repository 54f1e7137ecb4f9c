#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03q"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
for rep in 1 2; do
for q in default 8 2; do
  if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for fd in 1 0; do
    HN_BENCH_FORCE_DIST=$fd timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_q${q}_dist${fd}_$rep.json" 2>> "$out/bench.err" || exit 1
    echo "hw_queues=$q force_dist=$fd rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_q${q}_dist${fd}_$rep.json" | head -1)"
  done
done; done
