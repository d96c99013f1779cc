set -u
mkdir -p gpurun_out/r6c
python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_mirror.py tests/test_gpu_fused.py -m gpu -q > gpurun_out/r6c/pytest.txt 2>&1
tail -15 gpurun_out/r6c/pytest.txt
for i in 1 2 3; do
  tools/bin/exp_alloc_kind malloc 6 16 > gpurun_out/r6c/base_$i.txt 2>&1
  LD_PRELOAD=tools/bin/rot/libfastq_hip.so tools/bin/exp_alloc_kind malloc 6 16 > gpurun_out/r6c/rot_$i.txt 2>&1
done
grep -h "round 0" gpurun_out/r6c/base_*.txt | awk '{print "BASE", $0}'
grep -h "round 0" gpurun_out/r6c/rot_*.txt | awk '{print "ROT ", $0}'
