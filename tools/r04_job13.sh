mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_network.py -x -q -m gpu -k "stream or fullres or frozen" > gpurun_out/r4/t13.log 2>&1; tail -3 gpurun_out/r4/t13.log
L=$GRAFT_REPO_ROOT/heatnet-pub_amd/lib
NOTABLE=1 OLDENV="HEATNET_PREP_STREAM=0" bash tools/r04_ab.sh ab13 $L/libheatnet_hip.so
NOTABLE=1 ROUNDS=2 NEWENV="HN_BENCH_FORCE_DIST=1" OLDENV="HN_BENCH_FORCE_DIST=1 HEATNET_PREP_STREAM=0" bash tools/r04_ab.sh ab13d $L/libheatnet_hip.so
