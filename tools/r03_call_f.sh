#!/bin/bash
# Round-3 GPU collection, part F: streaming stem / head kernels and weight gradients on the side stream (A/B matrix), then the
# full GPU suite with both on.
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03f"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== tests $(date +%T)"
timeout -k 10 300 python3 -m pytest tests/test_gpu_small_convs.py -x -q > "$out/tests_small.log" 2>&1 || { tail -20 "$out/tests_small.log"; echo "small-kernel tests failed: stopping"; exit 1; }
tail -3 "$out/tests_small.log"
echo "== bench matrix $(date +%T)"
for cfg in "1 1" "0 1" "1 0" "0 0" "1 1"; do
  set -- $cfg
  HEATNET_SMALL_KERNELS=$1 HEATNET_WGRAD_STREAM=$2 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_small$1_ws$2.json" 2>> "$out/bench.err" || { echo "bench small=$1 ws=$2 failed"; tail -5 "$out/bench.err"; exit 1; }
  echo "small=$1 wgrad_stream=$2: $(grep -o '"value": [0-9.]*' "$out/bench_small$1_ws$2.json" | head -1)  $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_small$1_ws$2.json" | head -1)"
done
cp heatnet-pub_amd/tune_cache.json "$out/tune_cache.json"
HN_CONV_TABLE="$out/conv_table.txt" timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_tbl.json" 2>> "$out/bench.err"
echo "== full tests $(date +%T)"
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$out/pytest_gpu.log"; tail -6 "$out/pytest_gpu.log"
echo "== done $(date +%T)"
