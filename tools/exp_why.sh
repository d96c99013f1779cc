#!/bin/bash
# tools/exp_why.sh — tuning build; why does k_scan_stats2 give up a span?  (bits: 1 too many line starts, 2 no alignment, 4 a check failed, 8 alphabet, 32 tile)
cd "$(dirname "$0")/.."
mkdir -p /tmp/tuning
(cd fastq-rs_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFQH_TUNING -shared -o /tmp/tuning/libfastq_hip.so *.hip -ldl 2>&1 | grep -E "error")
FQH_FZ_WHY=1 FQH_LIB_PATH=/tmp/tuning/libfastq_hip.so python -m pytest tests/test_gpu_fused.py -x -q -k "sizes_around" 2>&1 | grep -E "FZ_WHY|passed|failed" | sort | uniq -c | tail
