#!/bin/bash
# Build container: copy the summaries of gpurun_out/final_r4 (tools/final_r04.sh a + b) into profiles/ under round-4 names.
src=gpurun_out/final_r4; dst=profiles
for w in seg confmax_seg confmax_critic seg_640x1408 seg_bf16 seg_bf16_640x1408 seg_rccl_rehearsal seg_resnext; do
  [ -f $src/bench_$w.json ] && tail -1 $src/bench_$w.json > $dst/r04_bench_$w.json
done
for w in seg confmax_seg confmax_critic seg_bf16 seg_resnext; do
  f=$(find $src/stats_$w -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $dst/r04_bench_${w}_kernel_stats.csv
done
tail -1 $src/stats_seg.json > $dst/r04_bench_seg_profiled.json
cp $src/conv_table_f16x2.txt $dst/r04_conv_table_f16x2.txt
cp $src/pmc_traffic_f16x2.json $src/pmc_traffic_f16x2.txt $src/pmc_traffic_bf16.json $src/pmc_traffic_bf16.txt $dst/
cp $src/smoke.txt $dst/r04_smoke.txt
cp $src/tune_cache.json heatnet-pub_amd/tune_cache.json
ls -la $dst | grep r04_ | wc -l
