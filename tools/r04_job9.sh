mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_planes.py tests/test_gpu_small_convs.py -x -q -m gpu > gpurun_out/r4/t8.log 2>&1; tail -3 gpurun_out/r4/t8.log
L=$GRAFT_REPO_ROOT/heatnet-pub_amd/lib
NOTABLE=1 OLDENV="HEATNET_WGRAD_HUGE=0" bash tools/r04_ab.sh ab8 $L/libheatnet_hip.so
NOTABLE=1 NEWENV="HEATNET_HIP_LIB=$L/libheatnet_hip_occ3.so" bash tools/r04_ab.sh ab9 $L/libheatnet_hip.so
