#!/bin/bash
# tools/exp_fz2time.sh [dbg flags..] — where a 4 KiB group's cycles go in k_scan_stats2 (tuning build with -DFQH_FZ_TIMING, made on the GPU box)
cd "$(dirname "$0")/.."
mkdir -p /tmp/tuning
(cd fastq-rs_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFQH_TUNING -DFQH_FZ_TIMING ${FQH_EXTRA_DEFS:-} -shared -o /tmp/tuning/libfastq_hip_t.so *.hip -ldl 2>&1 | grep -E "error")
for f in "${@:-0}"; do
FQH_FZ_DBG=$f FQH_LIB_PATH=/tmp/tuning/libfastq_hip_t.so python tools/exp_fzone.py 4 3 2>&1 | grep "FZ_TIMING\|kernel ms" | tail -2
done
