#!/bin/bash
# tools/exp_fz2.sh flags.. — knock-out timing of k_scan_stats2 (tuning build with -DFQH_TUNING made on the GPU box; results
# are wrong by design with any flag set: 1 no counting, 2 no record stage, 4 no per-entry pass)
cd "$(dirname "$0")/.."
mkdir -p /tmp/tuning
(cd fastq-rs_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFQH_TUNING ${FQH_EXTRA_DEFS:-} -shared -o /tmp/tuning/libfastq_hip.so *.hip -ldl 2>&1 | grep -E "error")
FQH_LIB_PATH=/tmp/tuning/libfastq_hip.so python tools/exp_fzdbg.py "$@"
