set -u
mkdir -p gpurun_out/r6i
python -m pytest tests -m gpu -x -q > gpurun_out/r6i/pytest.txt 2>&1
tail -12 gpurun_out/r6i/pytest.txt
python tools/fuzz_routes.py 120 631 2>&1 | tail -1 > gpurun_out/r6i/fuzz.txt
python tools/fuzz_streams.py 120 632 2>&1 | tail -1 >> gpurun_out/r6i/fuzz.txt
cat gpurun_out/r6i/fuzz.txt
timeout 900 python bench.py > gpurun_out/r6i/bench.json 2> gpurun_out/r6i/bench.err
python tools/bench_summary.py gpurun_out/r6i/bench.json
