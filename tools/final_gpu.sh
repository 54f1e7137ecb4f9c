#!/bin/bash
# Round-2 closing collection (GPU box, via gpurun, from the repo root): warm the tune cache for every workload, PMC traffic of
# the conv kernels per arithmetic, kernel statistics, SQ counters of the f16x2 kernel, smoke, the GPU suite, the final bench.
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/final_r2"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
step() { echo "== $1 $(date +%T)"; }
step "warm tune cache: all workloads"
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/warm_seg.json" 2> "$out/warm.err"
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload seg_bf16 > "$out/warm_bf16.json" 2>> "$out/warm.err"
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-other-workloads --workload confmax_seg > /dev/null 2>> "$out/warm.err"
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-other-workloads --workload confmax_critic > /dev/null 2>> "$out/warm.err"
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --height 640 --width 1408 > /dev/null 2>> "$out/warm.err"
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload seg_bf16 --height 640 --width 1408 > /dev/null 2>> "$out/warm.err"
cp heatnet-pub_amd/tune_cache.json "$out/tune_cache.json"
step "benches"
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads --workload confmax_seg > "$out/bench_confmax_seg.json" 2>> "$out/warm.err"
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads --workload confmax_critic > "$out/bench_confmax_critic.json" 2>> "$out/warm.err"
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --height 640 --width 1408 > "$out/bench_seg_640x1408.json" 2>> "$out/warm.err"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload seg_bf16 > "$out/bench_seg_bf16.json" 2>> "$out/warm.err"
HN_BENCH_FORCE_DIST=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_seg_rccl_rehearsal.json" 2>> "$out/warm.err"
HN_CONV_TABLE="$out/conv_table_f16x2.txt" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads > /dev/null 2>> "$out/warm.err"
step "kernel stats seg"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_seg" -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads > "$out/stats_seg.json" 2> "$out/stats_seg.err" || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
  step "pmc $ctr seg (f16x2)"
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/pmc_seg_$ctr" -- python3 bench.py --steps 2 --warmup 1 --workload seg --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/pmc_seg_$ctr.log" 2>&1 || exit 1
  step "pmc $ctr seg_bf16"
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/pmc_seg_bf16_$ctr" -- python3 bench.py --steps 2 --warmup 1 --workload seg_bf16 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/pmc_seg_bf16_$ctr.log" 2>&1 || exit 1
  step "pmc $ctr seg (bf16x3)"
  export HEATNET_CONV_MATH=bf16x3
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/pmc_seg_bf16x3_$ctr" -- python3 bench.py --steps 2 --warmup 1 --workload seg --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/pmc_seg_bf16x3_$ctr.log" 2>&1 || exit 1
  unset HEATNET_CONV_MATH
done
step "SQ counters, f16x2 forward kernel"
H2=1 bash tools/pmc_conv.sh 8,40,88,1024,2048,1,1,1 fwd 0 "$out/pmc_conv_h2" > "$out/pmc_conv_h2.log" 2>&1
step "smoke"
python3 -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1
step "full gpu tests"
python3 -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$out/pytest_gpu.log"
tail -4 "$out/pytest_gpu.log"
step "final bench"
python3 bench.py --steps 20 --warmup 5 > "$out/bench_seg.json" 2> "$out/bench_seg.err"
tail -c 400 "$out/bench_seg.json"
step "done"
