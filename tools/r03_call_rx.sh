#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03rx"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
timeout -k 10 300 python3 -m pytest tests/test_gpu_resnext.py -q -x > "$out/tests.log" 2>&1; tail -2 "$out/tests.log"
for rep in 1 2; do for du in 0 1; do
  echo "resnext dual=$du rep $rep: $(HEATNET_DUAL_STREAMS=$du timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --workload seg_resnext --no-cpu-baseline --no-alt --no-other-workloads --no-roofline 2>> "$out/bench.err" | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done; done
