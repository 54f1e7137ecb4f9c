#!/bin/bash
# Round-3 GPU collection, part B: the full GPU suite, then f16x2 on a trained state.
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
out="$root/gpurun_out/r03b"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root" || exit 1
echo "== tests $(date +%T)"
python3 -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$out/pytest_gpu.log"; tail -6 "$out/pytest_gpu.log"
echo "== trained state $(date +%T)"
timeout -k 10 500 python3 tools/trained_state_f16x2.py 2500 > "$out/trained_state.txt" 2>&1; tail -30 "$out/trained_state.txt"
echo "== done $(date +%T)"
