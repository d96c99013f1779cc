set -u
mkdir -p gpurun_out/r6f
for rep in 1 2; do
  tools/bin/exp_alloc_kind malloc 6 16 > gpurun_out/r6f/pol0_$rep.txt 2>&1
  for v in 1 2 3 4 5; do
    LD_PRELOAD=tools/bin/pol$v/libfastq_hip.so tools/bin/exp_alloc_kind malloc 6 16 > gpurun_out/r6f/pol${v}_$rep.txt 2>&1
  done
done
for v in 0 1 2 3 4 5; do echo "policy $v: $(grep -h 'round 1' gpurun_out/r6f/pol${v}_*.txt | sed 's/.*index kernel \([0-9.]*\) ms.*/\1/' | tr '\n' ' ')"; done
