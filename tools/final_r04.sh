#!/bin/bash
# Round-4 closing collection (GPU box, via gpurun, from the repo root; <= 20 min): tune cache for every workload, the benches,
# kernel statistics and PMC traffic of the headline step, smoke.   bash tools/final_r04.sh [part]   part = a | b
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/final_r4"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
step() { echo "== $1 $(date +%T)"; }
part=${1:-a}
commit=${HN_COMMIT:-unknown}
if [ "$part" = "a" ]; then
  step "warm tune cache"
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$out/warm_seg.json" 2> "$out/warm.err"
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --height 640 --width 1408 > /dev/null 2>> "$out/warm.err"
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload seg_bf16 --height 640 --width 1408 > /dev/null 2>> "$out/warm.err"
  cp heatnet-pub_amd/tune_cache.json "$out/tune_cache.json"
  step "benches"
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads --workload confmax_seg > "$out/bench_confmax_seg.json" 2>> "$out/warm.err"
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads --workload confmax_critic > "$out/bench_confmax_critic.json" 2>> "$out/warm.err"
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --height 640 --width 1408 > "$out/bench_seg_640x1408.json" 2>> "$out/warm.err"
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload seg_bf16 > "$out/bench_seg_bf16.json" 2>> "$out/warm.err"
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload seg_bf16 --height 640 --width 1408 > "$out/bench_seg_bf16_640x1408.json" 2>> "$out/warm.err"
  HN_BENCH_FORCE_DIST=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_seg_rccl_rehearsal.json" 2>> "$out/warm.err"
  HN_CONV_TABLE="$out/conv_table_f16x2.txt" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads > /dev/null 2>> "$out/warm.err"
  step "kernel stats seg"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_seg" -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads > "$out/stats_seg.json" 2> "$out/stats_seg.err"
  step "resnext"
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads --workload seg_resnext > "$out/bench_seg_resnext.json" 2>> "$out/warm.err"
  step "kernel stats confmax / bf16 / resnext"
  for wl in confmax_seg confmax_critic seg_bf16 seg_resnext; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$wl" -- python3 bench.py --steps 6 --warmup 2 --workload $wl --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/stats_$wl.json" 2> "$out/stats_$wl.err"
  done
  step "smoke"
  python3 -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1; tail -2 "$out/smoke.txt"
  python3 -c "import sys; sys.path.insert(0, \".\"); import bench" > /dev/null 2>&1
else
  for ctr in FETCH_SIZE WRITE_SIZE; do
    step "pmc $ctr seg (f16x2)"
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/pmc_seg_$ctr" -- python3 bench.py --steps 2 --warmup 1 --workload seg --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/pmc_seg_$ctr.log" 2>&1
    step "pmc $ctr seg_bf16"
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/pmc_seg_bf16_$ctr" -- python3 bench.py --steps 2 --warmup 1 --workload seg_bf16 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/pmc_seg_bf16_$ctr.log" 2>&1
  done
  ver=$(python3 -c "import sys; sys.path.insert(0, '.'); from heatnet_amd import _lib; print(_lib.load().hn_version())")
  python3 tools/pmc_aggregate.py "$out/pmc_seg_FETCH_SIZE" "$out/pmc_seg_WRITE_SIZE" "$out/pmc_traffic_f16x2" \
    "HBM traffic per launch, headline step (seg, bs 8, 320x704, f16x2 with plane tensors), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs" \
    "{\"workload\": \"seg\", \"batch\": 8, \"height\": 320, \"width\": 704, \"conv_math\": \"f16x2\", \"hn_version\": $ver, \"commit\": \"$commit\"}"
  python3 tools/pmc_aggregate.py "$out/pmc_seg_bf16_FETCH_SIZE" "$out/pmc_seg_bf16_WRITE_SIZE" "$out/pmc_traffic_bf16" \
    "HBM traffic per launch, seg_bf16 step (bs 8, 320x704, one-plane bf16 tensors)" \
    "{\"workload\": \"seg_bf16\", \"batch\": 8, \"height\": 320, \"width\": 704, \"conv_math\": \"bf16\", \"hn_version\": $ver, \"commit\": \"$commit\"}"
  head -30 "$out/pmc_traffic_f16x2.txt"
  step "final bench (with the traffic figures)"
  cp "$out/pmc_traffic_f16x2.json" profiles/pmc_traffic_f16x2.json; cp "$out/pmc_traffic_bf16.json" profiles/pmc_traffic_bf16.json
  python3 bench.py --steps 20 --warmup 5 > "$out/bench_seg.json" 2> "$out/bench_seg.err"
  tail -c 600 "$out/bench_seg.json"
fi
step "done"
