#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03w"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== tests $(date +%T)"
timeout -k 10 500 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py -q -x -k "leaky or confusion or critic or gradient_tensors" > "$out/tests.log" 2>&1 || { tail -60 "$out/tests.log"; exit 1; }
tail -2 "$out/tests.log"
for rep in 1 2; do
  for wl in confmax_critic confmax_seg; do
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --workload $wl --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_${wl}_$rep.json" 2>> "$out/bench.err" || exit 1
    echo "$wl rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_${wl}_$rep.json" | head -1)"
  done
done
echo "== done $(date +%T)"
