#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03l"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== tests $(date +%T)"
timeout -k 10 500 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py -q -x -k "conv2d_forward_backward or confusion or critic or backward_matches" > "$out/tests.log" 2>&1 || { tail -40 "$out/tests.log"; exit 1; }
tail -2 "$out/tests.log"
echo "== confmax $(date +%T)"
for wl in confmax_critic confmax_seg; do
  HN_CONV_TABLE="$out/conv_table_$wl.txt" timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --workload $wl --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_$wl.json" 2>> "$out/bench.err" || exit 1
  echo "$wl: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_$wl.json" | head -1)"
done
cp heatnet-pub_amd/tune_cache.json "$out/tune_cache.json"
echo "== done $(date +%T)"
