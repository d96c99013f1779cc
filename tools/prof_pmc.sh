#!/bin/bash
# tools/prof_pmc.sh "<counters pass 1>" "<counters pass 2>" ... — rocprofv3 PMC passes over a short
# bench run (run on the GPU box via gpurun).  Output under gpurun_out/prof_pmc/.  Every pass runs under
# its own timeout: a counter set the hardware cannot collect makes rocprofv3 abort and then hang.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_pmc
rm -rf $OUT; mkdir -p $OUT
CMD=${FQH_PROF_CMD:-"python bench.py --steps 2 --warmup 1 --no-cpu-baseline"}
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- $CMD > $OUT/pmc_$i.log 2>&1
done
python3 tools/prof_summary.py $OUT
