cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export HEATNET_CONV_MATH=f16x2 HEATNET_TUNE_CACHE=/tmp/tc_h2.json
python3 bench.py --steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline --no-alt > gpurun_out/b_h2.json 2> gpurun_out/b_h2.err || exit 1
python3 -c "
import sys; sys.path.insert(0,'.'); sys.argv=['x']
" 
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_h2 -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > gpurun_out/stats_h2.json 2> gpurun_out/stats_h2.err
