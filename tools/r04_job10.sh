mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r4/t10.log 2>&1; tail -3 gpurun_out/r4/t10.log
L=$GRAFT_REPO_ROOT/heatnet-pub_amd/lib
NOTABLE=1 bash tools/r04_ab.sh ab10 $L/libheatnet_hip_base.so
