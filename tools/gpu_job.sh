set -u
mkdir -p gpurun_out/r6e
python -m pytest tests -m gpu -x -q > gpurun_out/r6e/pytest.txt 2>&1
tail -12 gpurun_out/r6e/pytest.txt
{
echo "# fuzzers on round 6's code (differential against the oracle; tools/fuzz_*.py SECONDS SEED): external-source slots, chunks held and"
echo "# released out of order, parked rings in the ring fuzzer; a third of the sharded cases streamed in place (fqh_shard_stream_run_mapped)"
python tools/fuzz_streams.py 240 611 2>&1 | tail -1
python tools/fuzz_sharded.py 200 612 2>&1 | tail -1
python tools/fuzz_routes.py 150 613 2>&1 | tail -1
python tools/fuzz_protocol.py 100 614 2>&1 | tail -1
} > gpurun_out/r6e/fuzz.txt 2>&1
cat gpurun_out/r6e/fuzz.txt
timeout 900 python bench.py > gpurun_out/r6e/bench.json 2> gpurun_out/r6e/bench.err
python tools/bench_summary.py gpurun_out/r6e/bench.json
