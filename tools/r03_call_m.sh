#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03m"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== tests $(date +%T)"
timeout -k 10 300 python3 -m pytest tests/test_gpu_planes.py -q -k "bn_backward_reduction" > "$out/tests1.log" 2>&1 || { tail -60 "$out/tests1.log"; exit 1; }
tail -2 "$out/tests1.log"
if [ -n "$HN_NET_TESTS" ]; then
timeout -k 10 500 python3 -m pytest tests/test_gpu_network.py -q -x > "$out/tests2.log" 2>&1 || { tail -60 "$out/tests2.log"; exit 1; }
tail -2 "$out/tests2.log"
fi
echo "== A/B $(date +%T)"
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > /dev/null 2>> "$out/bench.err"
for rep in 1 2 3; do for fu in 0 1; do
  HEATNET_BN_REDUCE_FUSION=$fu timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_fuse${fu}_$rep.json" 2>> "$out/bench.err" || exit 1
  echo "fusion=$fu rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_fuse${fu}_$rep.json" | head -1)"
done; done
for fu in 0 1; do
  HEATNET_BN_REDUCE_FUSION=$fu timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --workload seg_bf16 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_bf16_fuse${fu}.json" 2>> "$out/bench.err" || exit 1
  echo "bf16 fusion=$fu: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_bf16_fuse${fu}.json" | head -1)"
done
cp heatnet-pub_amd/tune_cache.json "$out/tune_cache.json"
echo "== done $(date +%T)"
