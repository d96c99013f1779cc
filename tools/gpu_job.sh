set -u
mkdir -p gpurun_out/r6n
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r6n/pytest.txt 2>&1
tail -6 gpurun_out/r6n/pytest.txt
for rep in 1 2 3; do
  FQH_INDEX_WRITER=0 timeout 300 tools/bin/exp_alloc_kind malloc 6 16 > gpurun_out/r6n/w0_$rep.txt 2>&1
  FQH_INDEX_WRITER=1 timeout 300 tools/bin/exp_alloc_kind malloc 6 16 > gpurun_out/r6n/w1_$rep.txt 2>&1
done
echo "legacy: $(grep -h 'round 1' gpurun_out/r6n/w0_*.txt | sed 's/.*index kernel \([0-9.]*\) ms.*/\1/' | tr '\n' ' ')"
echo "writer: $(grep -h 'round 1' gpurun_out/r6n/w1_*.txt | sed 's/.*index kernel \([0-9.]*\) ms.*/\1/' | tr '\n' ' ')"
tail -2 gpurun_out/r6n/w1_1.txt
