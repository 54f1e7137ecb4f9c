#!/bin/bash
# Quick look at one configuration (GPU box, via gpurun, from the repo root): a timed bench line, then rocprofv3 kernel statistics
# of the same step.   [HEATNET_CONV_MATH=...] bash tools/prof_step.sh [tag]
tag=${1:-step}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export HEATNET_TUNE_CACHE=${HEATNET_TUNE_CACHE:-/tmp/tc_$tag.json}
python3 bench.py --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline --no-alt > gpurun_out/b_$tag.json 2> gpurun_out/b_$tag.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_$tag -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > gpurun_out/stats_$tag.json 2> gpurun_out/stats_$tag.err
