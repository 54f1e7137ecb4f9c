#!/bin/bash
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03z"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
cp heatnet-pub_amd/tune_cache.json "$out/tune_cache_plain.json"
echo "== baseline (committed cache) $(date +%T)"
for rep in 1 2; do
  timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_plain_$rep.json" 2>> "$out/bench.err" || exit 1
  echo "plain rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_plain_$rep.json" | head -1)"
done
echo "== re-tune with cache flushes $(date +%T)"
export HEATNET_TUNE_CACHE="$out/tune_cache_flush.json"
HEATNET_TUNE_FLUSH=1 timeout -k 10 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_tune.json" 2>> "$out/bench.err" || exit 1
for rep in 1 2; do
  timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-other-workloads --no-roofline > "$out/bench_flush_$rep.json" 2>> "$out/bench.err" || exit 1
  echo "flush-tuned rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/bench_flush_$rep.json" | head -1)"
done
HN_CONV_TABLE="$out/conv_table_flush.txt" timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads > /dev/null 2>> "$out/bench.err"
unset HEATNET_TUNE_CACHE
HN_CONV_TABLE="$out/conv_table_plain.txt" timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-workloads > /dev/null 2>> "$out/bench.err"
echo "== done $(date +%T)"
