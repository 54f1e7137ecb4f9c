#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03dual"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
for rep in 1 2 3; do for du in 0 1; do
  HEATNET_DUAL_STREAMS=$du timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_dual${du}_$rep.json" 2>> "$out/bench.err" || exit 1
  echo "dual=$du rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_dual${du}_$rep.json" | head -1)"
done; done
HEATNET_DUAL_STREAMS=1 timeout -k 10 400 python3 -m pytest tests/test_gpu_network.py -q -x -k "backward_matches or forward_matches_reference or gradient_tensors" > "$out/tests.log" 2>&1; tail -3 "$out/tests.log"
