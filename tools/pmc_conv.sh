#!/bin/bash
# PMC passes over one conv microbenchmark shape (run on the GPU box, from the repo root):
#   [H2=1] bash tools/pmc_conv.sh <shape> <kind> <variant> <outdir>      (H2=1: the f16x2 forward kernel)
# Each pass is its own rocprofv3 run with --kernel-trace only (counter groups limited by the per-pass slots).
shape=${1:-8,40,88,1024,2048,1,1,1}; kind=${2:-fwd}; variant=${3:-128}; out=${4:-gpurun_out/pmc_conv}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for group in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_CYCLES" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
             "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
             "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  H2=${H2:-0} KINDS=$kind VARIANT=$variant rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$out/p$i" -- python3 tools/bench_conv.py "$shape" > "$out.p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out.p$i.log"; }
done
