mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pyramid or avgpool or adaptive" > gpurun_out/r4/t12.log 2>&1; tail -3 gpurun_out/r4/t12.log
L=$GRAFT_REPO_ROOT/heatnet-pub_amd/lib
NOTABLE=1 OLDENV="HEATNET_PYRAMID_FWD_FUSED=0" bash tools/r04_ab.sh ab12 $L/libheatnet_hip.so
