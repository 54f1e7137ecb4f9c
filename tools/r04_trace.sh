#!/bin/bash
# GPU box: kernel trace (start / end of every dispatch) of a few headline steps -> tools/trace_gaps.py
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r4/trace"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > /dev/null 2> "$out/err.log" || exit 1
rocprofv3 --kernel-trace --output-format csv -d "$out/t" -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench.json" 2>> "$out/err.log" || exit 1
f=$(find "$out/t" -name '*kernel_trace.csv' | head -1)
python3 tools/trace_gaps.py "$f" > "$out/gaps.txt" 2>&1
rm -rf "$out/t"
cat "$out/gaps.txt"
