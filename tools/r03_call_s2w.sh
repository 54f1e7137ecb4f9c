#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03s2w"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
b() { timeout -k 10 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline "$@" 2>> "$out/bench.err" | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for rep in 1 2; do for v in 0 1; do
  echo "second_stream_wgrad=$v rep $rep: seg $(HEATNET_SECOND_STREAM_WGRAD=$v b)  dist $(HN_BENCH_FORCE_DIST=1 HEATNET_SECOND_STREAM_WGRAD=$v b)  bf16 $(HEATNET_SECOND_STREAM_WGRAD=$v b --workload seg_bf16)  confmax_seg $(HEATNET_SECOND_STREAM_WGRAD=$v b --workload confmax_seg)"
done; done
HEATNET_SECOND_STREAM_WGRAD=1 timeout -k 10 400 python3 -m pytest tests/test_gpu_network.py tests/test_gpu_ddp.py -q -x -k "backward_matches or two_encoder or confusion or two_ranks" > "$out/tests.log" 2>&1; tail -2 "$out/tests.log"
