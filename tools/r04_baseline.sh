#!/bin/bash
# Round-4 baseline (GPU box): headline bench, per-shape conv table, one-stream kernel statistics.
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r4/${1:-base}"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads > "$out/bench_seg.json" 2> "$out/bench.err" || exit 1
tail -c 400 "$out/bench_seg.json"; echo
HN_CONV_TABLE="$out/conv_table.txt" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads > /dev/null 2>> "$out/bench.err" || exit 1
HEATNET_DUAL_STREAMS=0 HEATNET_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_one_stream" -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/stats_one_stream.json" 2> "$out/stats.err" || exit 1
find "$out/stats_one_stream" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats_one_stream.csv" \;
find "$out/stats_one_stream" -name '*.csv' ! -name '*kernel_stats.csv' -delete
tail -c 300 "$out/stats_one_stream.json"; echo; echo done
