#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r4"; mkdir -p "$out"
cd "$root" || exit 1
export HEATNET_HIP_LIB=$root/heatnet-pub_amd/lib/stamps/libheatnet_hip_stamps.so
{
for t in 2 7; do python3 tools/stamp_conv.py fwd 8,40,88,2048,512,1,1,1 $t || exit 1; done
for t in 2 7; do python3 tools/stamp_conv.py fwd 8,80,176,1024,256,3,1,1 $t || exit 1; done
} > "$out/${1:-stamps}.txt" 2>&1
grep -E "^(fwd|dgrad)|K loop|step 3|per workgroup" "$out/${1:-stamps}.txt"
