#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03g"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
python3 -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
echo "== priority experiments $(date +%T)"
for cfg in "0 0" "1 0" "0 0" "1 0"; do
  set -- $cfg
  HEATNET_MAIN_HIGH=$1 HEATNET_SIDE_PRIORITY=$2 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_mh$1_sp$2.json" 2>> "$out/bench.err"
  echo "main_high=$1 side_prio=$2: $(grep -o '"value": [0-9.]*' "$out/bench_mh$1_sp$2.json" | head -1)  $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_mh$1_sp$2.json" | head -1)"
done
echo "== full tests $(date +%T)"
timeout -k 10 900 python3 -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$out/pytest_gpu.log"; tail -8 "$out/pytest_gpu.log"
echo "== done $(date +%T)"
