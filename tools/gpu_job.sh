set -u
mkdir -p gpurun_out/r6l
for rep in 1 2 3; do
  tools/bin/exp_alloc_kind malloc 6 16 > gpurun_out/r6l/base_$rep.txt 2>&1
  FQH_EXP_IGNORE=1 LD_PRELOAD=tools/bin/tr/libfastq_hip.so tools/bin/exp_alloc_kind malloc 6 16 > gpurun_out/r6l/tr_$rep.txt 2>&1
done
echo "base  with-stores (first 4 GiB, fresh line buffer): $(grep -h 'round 0' gpurun_out/r6l/base_*.txt | sed 's/.* with \([0-9.]*\)$/\1/' | tr '\n' ' ')"
echo "base  without:                                      $(grep -h 'round 0' gpurun_out/r6l/base_*.txt | sed 's/.*without stores \([0-9.]*\),.*/\1/' | tr '\n' ' ')"
echo "trans with-stores:                                  $(grep -h 'round 0' gpurun_out/r6l/tr_*.txt | sed 's/.* with \([0-9.]*\)$/\1/' | tr '\n' ' ')"
echo "trans without:                                      $(grep -h 'round 0' gpurun_out/r6l/tr_*.txt | sed 's/.*without stores \([0-9.]*\),.*/\1/' | tr '\n' ' ')"
tail -2 gpurun_out/r6l/tr_1.txt
