#!/bin/bash
# tools/build_variant.sh NAME [SRC_OF_fused_kernels.hip] [extra hipcc flags] — a copy of the library with another
# fused_kernels object (tools/bin/NAME.so; git-ignored, travels with gpurun), for tools/ab_stats.py: builds A / B of
# k_scan_stats side by side in one process.  The other objects come from fastq-rs_amd/csrc/build (run make there first).
set -eu
cd "$(dirname "$0")/.."
NAME=$1; SRC=${2:-fastq-rs_amd/csrc/fused_kernels.hip}; shift; shift || true
mkdir -p tools/bin/obj
cp "$SRC" tools/bin/obj/fused_$NAME.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Ifastq-rs_amd/csrc "$@" -c -o tools/bin/obj/fused_$NAME.o tools/bin/obj/fused_$NAME.hip
OBJS=$(ls fastq-rs_amd/csrc/build/*.o | grep -v fused_kernels)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/$NAME.so $OBJS tools/bin/obj/fused_$NAME.o -ldl
echo built tools/bin/$NAME.so
