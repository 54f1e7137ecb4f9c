#!/bin/bash
# GPU box: first tuning pass (isolated launches), headline; in-situ pass; headline again, alternating with the first cache.
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r4/insitu"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline"
val() { python3 -c "import json,sys; b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], b['value'], b['ms_per_step'])" "$1"; }
HEATNET_TUNE_CACHE=$out/tune_a.json $B > $out/a0.json 2> $out/err.log || exit 1
cp $out/tune_a.json $out/tune_b.json
HEATNET_TUNE_CACHE=$out/tune_b.json python3 tools/insitu_tune.py --top ${TOP:-60} > $out/insitu.log 2>> $out/err.log || { tail -5 $out/err.log; exit 1; }
tail -25 $out/insitu.log
for i in 1 2 3; do
  HEATNET_TUNE_CACHE=$out/tune_a.json $B > $out/a$i.json 2>> $out/err.log || exit 1; val $out/a$i.json
  HEATNET_TUNE_CACHE=$out/tune_b.json $B > $out/b$i.json 2>> $out/err.log || exit 1; val $out/b$i.json
done
echo done
