#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03t"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== tests $(date +%T)"
timeout -k 10 400 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_resnext.py tests/test_gpu_network.py -q -x -k "conv2d_forward_backward or resnext or custom or feedback" > "$out/tests.log" 2>&1 || { tail -60 "$out/tests.log"; exit 1; }
tail -2 "$out/tests.log"
echo "== resnext $(date +%T)"
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --workload seg_resnext --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_seg_resnext.json" 2>> "$out/bench.err" || exit 1
echo "seg_resnext: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_seg_resnext.json" | head -1)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_seg_resnext" -- python3 bench.py --steps 6 --warmup 2 --workload seg_resnext --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/stats_seg_resnext.json" 2> "$out/stats_seg_resnext.err"
cp heatnet-pub_amd/tune_cache.json "$out/tune_cache.json"
echo "== done $(date +%T)"
