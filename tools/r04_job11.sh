mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pyramid or bilinear or cross_entropy or ce_" > gpurun_out/r4/t11.log 2>&1; tail -3 gpurun_out/r4/t11.log
L=$GRAFT_REPO_ROOT/heatnet-pub_amd/lib
NOTABLE=1 OLDENV="HEATNET_PYRAMID_BWD_FUSED=0" bash tools/r04_ab.sh ab11 $L/libheatnet_hip.so
