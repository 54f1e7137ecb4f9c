#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03dual3"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
b() { timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline 2>> "$out/bench.err" | grep -o '"ms_per_step": [0-9.]*' | head -1; }
export HN_BENCH_FORCE_DIST=1
echo "dist dual=1 baseline: $(HEATNET_DUAL_STREAMS=1 b)"
echo "dist dual=1 no wgrad side stream: $(HEATNET_DUAL_STREAMS=1 HEATNET_WGRAD_STREAM=0 b)"
echo "dist dual=1 16 hw queues: $(HEATNET_DUAL_STREAMS=1 GPU_MAX_HW_QUEUES=16 b)"
echo "dist dual=1 without the extra comm waits (UNSAFE, diagnosis): $(HEATNET_DUAL_STREAMS=1 HN_DDP_NO_DUAL_WAIT=1 b)"
echo "dist dual=1 main high: $(HEATNET_DUAL_STREAMS=1 HEATNET_MAIN_HIGH=1 b)"
echo "dist dual=0: $(HEATNET_DUAL_STREAMS=0 b)"
timeout -k 10 300 python3 -m pytest tests/test_gpu_network.py -q -x -k "two_encoder" > "$out/tests.log" 2>&1; tail -2 "$out/tests.log"
