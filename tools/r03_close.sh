#!/bin/bash
# closing run of round 3 at HEAD: full GPU suite, the default bench line, the other workloads, smoke
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03_close"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== full suite $(date +%T)"
timeout -k 10 900 python3 -m pytest tests -m gpu -q > "$out/tests_full.log" 2>&1; rc=$?
grep -E "^FAILED|passed|failed" "$out/tests_full.log" | tail -8
echo "== benches $(date +%T)"
python3 bench.py > "$out/bench_seg.json" 2> "$out/bench_seg.err"; echo "seg: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_seg.json" | head -1)"
b() { python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads "$@" 2>> "$out/bench.err"; }
b --workload confmax_seg > "$out/bench_confmax_seg.json"; echo "confmax_seg: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_confmax_seg.json" | head -1)"
b --workload confmax_critic > "$out/bench_confmax_critic.json"; echo "confmax_critic: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_confmax_critic.json" | head -1)"
b --workload seg_bf16 > "$out/bench_seg_bf16.json"; echo "seg_bf16: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_seg_bf16.json" | head -1)"
b --height 640 --width 1408 > "$out/bench_seg_640x1408.json"; echo "640x1408: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_seg_640x1408.json" | head -1)"
b --workload seg_bf16 --height 640 --width 1408 > "$out/bench_seg_bf16_640x1408.json"; echo "bf16 640x1408: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_seg_bf16_640x1408.json" | head -1)"
b --workload seg_resnext > "$out/bench_seg_resnext.json"; echo "resnext: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_seg_resnext.json" | head -1)"
HN_BENCH_FORCE_DIST=1 b > "$out/bench_seg_rccl_rehearsal.json"; echo "N>1 path: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_seg_rccl_rehearsal.json" | head -1)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_seg" -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads > "$out/stats_seg.json" 2> "$out/stats_seg.err"
python3 -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.txt" 2>&1; tail -1 "$out/smoke.txt"
cp heatnet-pub_amd/tune_cache.json "$out/tune_cache.json"
echo "== done rc=$rc $(date +%T)"
exit $rc
