#!/bin/bash
# tools/build_tuning.sh [NAME] [extra hipcc flags] — a -DFQH_TUNING build of the whole library (tools/bin/NAME.so, default
# tune.so; git-ignored, travels with gpurun): the experiment hooks of csrc/ (FQH_TUNE_LINES_OFFSET, knock-outs, ...) are in,
# the symbol table is open.  Built HERE (hipcc cross-compiles), so that no GPU-minute is spent compiling.
set -eu
cd "$(dirname "$0")/.."
NAME=${1:-tune}; shift || true
mkdir -p tools/bin/obj_$NAME
pids=()
for f in fastq-rs_amd/csrc/*.hip; do
  b=$(basename "$f" .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFQH_TUNING "$@" -c -o tools/bin/obj_$NAME/$b.o "$f" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/$NAME.so tools/bin/obj_$NAME/*.o -ldl
echo built tools/bin/$NAME.so
