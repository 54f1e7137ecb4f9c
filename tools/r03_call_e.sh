#!/bin/bash
# Round-3 GPU collection, part E: weight gradients on the side stream (A/B), then the full GPU suite with it on.
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03e"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== bench A/B $(date +%T)"
for rep in 1 2; do
  HEATNET_WGRAD_STREAM=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_off_$rep.json" 2>> "$out/bench.err"
  HEATNET_WGRAD_STREAM=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_on_$rep.json" 2>> "$out/bench.err"
done
grep -o '"value": [0-9.]*, "unit": "images/s", "n_gpus"' "$out"/bench_o*.json
HN_BENCH_FORCE_DIST=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_rccl_rehearsal.json" 2>> "$out/bench.err"; tail -c 900 "$out/bench_rccl_rehearsal.json"
echo "== tests $(date +%T)"
python3 -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$out/pytest_gpu.log"; tail -6 "$out/pytest_gpu.log"
echo "== done $(date +%T)"
