#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03p"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
for side in 1 0; do
  HEATNET_WGRAD_STREAM=$side timeout -k 10 200 python3 tools/host_time.py seg 2>/dev/null | tail -1
  HEATNET_WGRAD_STREAM=$side HN_HOST_TIME_DIST=1 timeout -k 10 200 python3 tools/host_time.py seg 2>/dev/null | tail -1
done
HN_HOST_TIME_DIST=1 timeout -k 10 300 python3 -c "
import cProfile, pstats, sys, os
sys.argv=['host_time.py','seg']
sys.path.insert(0,'tools')
import host_time
cProfile.run('host_time.main()', '$out/prof.bin')
p=pstats.Stats('$out/prof.bin'); p.sort_stats('tottime').print_stats(28)
" > "$out/cprofile_dist.txt" 2>&1
tail -45 "$out/cprofile_dist.txt"
