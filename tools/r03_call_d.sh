#!/bin/bash
# Round-3 GPU collection, part D: streaming kernels for stems / head (tests, A/B), headline bench, per-shape table, kernel
# statistics of the confusion-maximisation iterations.
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03d"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== tests $(date +%T)"
python3 -m pytest tests/test_gpu_small_convs.py tests/test_gpu_planes.py -x -q > "$out/tests.log" 2>&1; tail -5 "$out/tests.log"
python3 -m pytest tests/test_gpu_network.py -x -q -k "fixture or rgb18 or fullres" > "$out/tests_net.log" 2>&1; tail -3 "$out/tests_net.log"
echo "== bench A/B $(date +%T)"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_small_on.json" 2> "$out/bench.err"; tail -c 200 "$out/bench_small_on.json"
HEATNET_SMALL_KERNELS=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_small_off.json" 2>> "$out/bench.err"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_small_on2.json" 2>> "$out/bench.err"
cp heatnet-pub_amd/tune_cache.json "$out/tune_cache.json"
echo "== conv table $(date +%T)"
HN_CONV_TABLE="$out/conv_table.txt" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_tbl.json" 2>> "$out/bench.err"
echo "== confmax stats $(date +%T)"
for wl in confmax_critic confmax_seg; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$wl" -- python3 bench.py --steps 6 --warmup 2 --workload $wl --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/stats_$wl.json" 2> "$out/stats_$wl.err"
done
echo "== done $(date +%T)"
