#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r4"; mkdir -p "$out"
cd "$root" || exit 1
export HEATNET_HIP_LIB=$root/heatnet-pub_amd/lib/stamps/libheatnet_hip_stamps.so
{
for sh in 8,80,176,64,256,1,1,1 8,40,88,128,512,1,1,1 8,40,88,256,1024,1,1,1 8,40,88,1024,256,1,1,1 8,80,176,256,64,1,1,1; do
  for t in 0 5; do python3 tools/stamp_conv.py fwd $sh $t || exit 1; done
done
python3 tools/stamp_conv.py dgrad 8,40,88,512,128,1,1,1 0 || exit 1
python3 tools/stamp_conv.py fwd 8,40,88,512,512,3,1,4 0 || exit 1
python3 tools/stamp_conv.py fwd 8,40,88,512,2048,1,1,1 2 || exit 1
} > "$out/${1:-stamps}.txt" 2>&1
tail -3 "$out/${1:-stamps}.txt"
