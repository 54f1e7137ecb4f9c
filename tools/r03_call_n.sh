#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03n"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== tests $(date +%T)"
timeout -k 10 400 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_resnext.py tests/test_gpu_planes.py -q -x -k "conv2d_forward_backward or resnext or custom or bn_backward_reduction" > "$out/tests.log" 2>&1 || { tail -60 "$out/tests.log"; exit 1; }
tail -2 "$out/tests.log"
echo "== resnext $(date +%T)"
for pc in 0 1; do
  HEATNET_PAD_CIN=$pc timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --workload seg_resnext --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_resnext_pad$pc.json" 2>> "$out/bench.err" || exit 1
  echo "pad_cin=$pc: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_resnext_pad$pc.json" | head -1)"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_seg_resnext" -- python3 bench.py --steps 6 --warmup 2 --workload seg_resnext --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/stats_seg_resnext.json" 2> "$out/stats_seg_resnext.err"
echo "== confmax_seg $(date +%T)"
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --workload confmax_seg --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_confmax_seg.json" 2>> "$out/bench.err" || exit 1
echo "confmax_seg: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_confmax_seg.json" | head -1) $(grep -o '"bn_backward_reductions": {[^}]*}' "$out/bench_confmax_seg.json")"
echo "== seg without side stream, kernel stats $(date +%T)"
HEATNET_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_seg_noside" -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads > "$out/stats_seg_noside.json" 2> "$out/stats_seg_noside.err"
echo "noside: $(grep -o '"ms_per_step": [0-9.]*' "$out/stats_seg_noside.json" | head -1)"
cp heatnet-pub_amd/tune_cache.json "$out/tune_cache.json"
echo "== done $(date +%T)"
