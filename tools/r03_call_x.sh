#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03x"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
for rep in 1 2; do for ps in 0 1; do
  for wl in confmax_critic confmax_seg; do
    HEATNET_PAD_SMALL_CIN=$ps timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --workload $wl --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_${wl}_ps${ps}_$rep.json" 2>> "$out/bench.err" || exit 1
    echo "pad_small=$ps $wl rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_${wl}_ps${ps}_$rep.json" | head -1)"
  done
done; done
HEATNET_PAD_SMALL_CIN=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_network.py -q -x -k "confusion or gradient_tensors" > "$out/tests.log" 2>&1; tail -2 "$out/tests.log"
