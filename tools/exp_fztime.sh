#!/bin/bash
# tools/exp_fztime.sh — builds a tuning copy of the library with -DFQH_FZ_TIMING (k_scan_stats adds up the cycles a
# wavefront spends in each phase of a 4 KiB group) into gpurun_out/tuning/ and runs 4 GiB launches with it.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/tuning
(cd fastq-rs_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFQH_FZ_TIMING ${FQH_EXTRA_DEFS:-} -shared -o ../../gpurun_out/tuning/libfastq_hip.so *.hip -ldl 2>&1 | grep -E "error")
FQH_LIB_PATH=$PWD/gpurun_out/tuning/libfastq_hip.so python tools/exp_fzone.py 4 3 2>&1 | grep "FZ_TIMING\|kernel ms" | tail -4
