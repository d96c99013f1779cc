set -u
mkdir -p gpurun_out/r6g
for lo in 0 149 100 36; do
  for fused in 1 0; do
    echo "== RAGGED=$lo FQH_FUSED=$fused" >> gpurun_out/r6g/ragged.txt
    RAGGED=$lo FQH_FUSED=$fused python tools/sweep_read_length.py 8 150 2>&1 | grep "read length" >> gpurun_out/r6g/ragged.txt
  done
done
for lo in 100 36; do
  for fused in 1 0; do
    echo "== RAGGED=$lo L=100 FQH_FUSED=$fused" >> gpurun_out/r6g/ragged.txt
    RAGGED=$lo FQH_FUSED=$fused python tools/sweep_read_length.py 8 100 250 2>&1 | grep "read length" >> gpurun_out/r6g/ragged.txt
  done
done
cat gpurun_out/r6g/ragged.txt | cut -c1-200
