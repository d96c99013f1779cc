set -u
mkdir -p gpurun_out/r6j
python -m pytest tests -m gpu -q > gpurun_out/r6j/pytest.txt 2>&1
tail -25 gpurun_out/r6j/pytest.txt
