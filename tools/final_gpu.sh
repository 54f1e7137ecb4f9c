#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/final_r2"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
step() { echo "== $1 $(date +%T)"; }
step "warm tune cache: all workloads"
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/warm_seg.json" 2> "$out/warm.err"
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload seg_bf16 > "$out/warm_bf16.json" 2>> "$out/warm.err"
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-other-workloads --workload confmax_seg > /dev/null 2>> "$out/warm.err"
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-other-workloads --workload confmax_critic > /dev/null 2>> "$out/warm.err"
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-other-workloads --height 640 --width 1408 > /dev/null 2>> "$out/warm.err"
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload seg_bf16 --height 640 --width 1408 > /dev/null 2>> "$out/warm.err"
cp heatnet-pub_amd/tune_cache.json "$out/tune_cache.json"
step "train curves"
python3 tools/train_curve.py 30 > "$out/train_curves.txt" 2>> "$out/warm.err"
for wl in seg seg_bf16; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    step "pmc $ctr $wl"
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/pmc_${wl}_$ctr" -- python3 bench.py --steps 2 --warmup 1 --workload $wl --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/pmc_${wl}_$ctr.log" 2>&1 || exit 1
  done
done
step "smoke"
python3 -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1
step "full gpu tests"
python3 -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$out/pytest_gpu.log"
tail -4 "$out/pytest_gpu.log"
step "final bench"
python3 bench.py --steps 20 --warmup 5 > "$out/bench_seg.json" 2> "$out/bench_seg.err"
tail -c 400 "$out/bench_seg.json"
step "done"
