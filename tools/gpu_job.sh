set -u
mkdir -p gpurun_out/r6d
python -m pytest tests -m gpu -x -q > gpurun_out/r6d/pytest.txt 2>&1
tail -25 gpurun_out/r6d/pytest.txt
timeout 900 python bench.py > gpurun_out/r6d/bench.json 2> gpurun_out/r6d/bench.err
tail -c 300 gpurun_out/r6d/bench.err
