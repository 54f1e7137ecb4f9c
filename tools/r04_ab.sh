#!/bin/bash
# A/B of two builds of the kernel library on one box (GPU box): alternating headline runs, then per-shape conv tables.
#   bash tools/r04_ab.sh <tag> [old_lib]     old_lib defaults to heatnet-pub_amd/lib/libheatnet_hip_r3.so
#   NEWENV="VAR=val ..." / OLDENV="..." apply to the runs of the new / old library, ROUNDS (default 3) = alternations
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
tag=${1:-ab}; old=${2:-$root/heatnet-pub_amd/lib/libheatnet_hip_r3.so}
out="$root/gpurun_out/r4/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline"
val() { python3 -c "import json,sys; b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], b['value'], b['ms_per_step'])" "$1"; }
new() { env $NEWENV HEATNET_TUNE_CACHE=$out/tune_new.json "$@"; }
oldr() { env $OLDENV HEATNET_HIP_LIB=$old HEATNET_TUNE_CACHE=$out/tune_old.json "$@"; }
new $B > $out/new0.json 2> $out/err.log || exit 1
oldr $B > $out/old0.json 2>> $out/err.log || exit 1
for i in $(seq 1 ${ROUNDS:-3}); do
  new $B > $out/new$i.json 2>> $out/err.log || exit 1; val $out/new$i.json
  oldr $B > $out/old$i.json 2>> $out/err.log || exit 1; val $out/old$i.json
done
if [ -z "$NOTABLE" ]; then
HN_CONV_TABLE="$out/conv_table_new.txt" new python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads > $out/roof_new.json 2>> $out/err.log || exit 1
HN_CONV_TABLE="$out/conv_table_old.txt" oldr python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads > $out/roof_old.json 2>> $out/err.log || exit 1
fi
echo done
