#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03dual4"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
b() { timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline "$@" 2>> "$out/bench.err" | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for rep in 1 2; do
for cfg in "1 1" "1 0" "0 1"; do
  set -- $cfg
  echo "plain dual=$1 side=$2 rep $rep: $(HEATNET_DUAL_STREAMS=$1 HEATNET_WGRAD_STREAM=$2 b)   dist: $(HN_BENCH_FORCE_DIST=1 HEATNET_DUAL_STREAMS=$1 HEATNET_WGRAD_STREAM=$2 b)   bf16: $(HEATNET_DUAL_STREAMS=$1 HEATNET_WGRAD_STREAM=$2 b --workload seg_bf16)  confmax_seg: $(HEATNET_DUAL_STREAMS=$1 HEATNET_WGRAD_STREAM=$2 b --workload confmax_seg)"
done; done
timeout -k 10 300 python3 -m pytest tests/test_gpu_network.py -q -x -k "two_encoder" > "$out/tests.log" 2>&1; tail -2 "$out/tests.log"
