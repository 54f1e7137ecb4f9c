#!/bin/bash
# Round-3 GPU collection, part C: the one-plane bf16 arithmetic (kernel tests, network tests, bench with / without plane
# tensors), then f16x2 on a trained state (second version of the tool).
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03c"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== tests $(date +%T)"
python3 -m pytest tests/test_gpu_planes.py -x -q > "$out/tests_planes.log" 2>&1; tail -5 "$out/tests_planes.log"
python3 -m pytest tests/test_gpu_network.py tests/test_gpu_kernels.py -x -q -k "bf16" > "$out/tests_bf16.log" 2>&1; tail -5 "$out/tests_bf16.log"
echo "== bench seg_bf16 $(date +%T)"
python3 bench.py --steps 20 --warmup 5 --workload seg_bf16 --no-cpu-baseline > "$out/bench_seg_bf16.json" 2> "$out/bench_seg_bf16.err"; tail -c 700 "$out/bench_seg_bf16.json"
HEATNET_PLANES=0 python3 bench.py --steps 20 --warmup 5 --workload seg_bf16 --no-cpu-baseline > "$out/bench_seg_bf16_noplanes.json" 2>> "$out/bench_seg_bf16.err"; tail -c 300 "$out/bench_seg_bf16_noplanes.json"
cp heatnet-pub_amd/tune_cache.json "$out/tune_cache.json"
echo "== trained state $(date +%T)"
timeout -k 10 560 python3 tools/trained_state_f16x2.py 3500 > "$out/trained_state.txt" 2>&1; tail -32 "$out/trained_state.txt"
echo "== done $(date +%T)"
