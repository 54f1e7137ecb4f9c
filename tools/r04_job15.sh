mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_planes.py -x -q -m gpu > gpurun_out/r4/t15.log 2>&1; tail -3 gpurun_out/r4/t15.log
L=$GRAFT_REPO_ROOT/heatnet-pub_amd/lib
OLDENV="HEATNET_ROW7_TILE=0" bash tools/r04_ab.sh ab15 $L/libheatnet_hip.so
