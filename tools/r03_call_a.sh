#!/bin/bash
# Round-3 GPU collection, part A (gpurun, from the repo root): new-kernel tests, microbenchmarks of the deep-prefetch tiles, the
# default bench (fills the tune cache), per-shape table, kernel statistics.
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03a"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
step() { echo "== $1 $(date +%T)"; }
step "tests"
python3 -m pytest tests/test_gpu_planes.py tests/test_gpu_kernels.py -x -q -k "planes or overlap or guard or fused or round" > "$out/tests.log" 2>&1; tail -4 "$out/tests.log"
step "microbench"
timeout -k 10 300 python3 tools/bench_planes.py 8,40,88,1024,256,1,1,1 8,40,88,256,1024,1,1,1 8,80,176,64,256,1,1,1 8,80,176,256,64,1,1,1 8,40,88,128,512,1,1,1 8,40,88,512,128,1,1,1 8,80,176,64,64,3,1,1 8,40,88,128,128,3,1,1 8,160,352,256,64,3,1,1 > "$out/bench_planes.log" 2>&1
step "bench (default)"
python3 bench.py --steps 20 --warmup 5 > "$out/bench_seg.json" 2> "$out/bench_seg.err"
cp heatnet-pub_amd/tune_cache.json "$out/tune_cache.json"
step "conv table"
HN_CONV_TABLE="$out/conv_table.txt" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_tbl.json" 2>> "$out/bench_seg.err"
step "kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_seg" -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/stats_seg.json" 2> "$out/stats_seg.err"
step "done"
