mkdir -p gpurun_out/r4
HN_TAP_INNER=1 python tools/probes/tap_order_check.py > gpurun_out/r4/tap_check.txt 2>&1 || { tail -5 gpurun_out/r4/tap_check.txt; exit 1; }
cat gpurun_out/r4/tap_check.txt
L=$GRAFT_REPO_ROOT/heatnet-pub_amd/lib
NEWENV="HN_TAP_INNER=1" OLDENV="HN_TAP_INNER=0" bash tools/r04_ab.sh abtap $L/libheatnet_hip.so
