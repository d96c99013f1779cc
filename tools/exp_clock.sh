#!/bin/bash
# samples rocm-smi clocks/power while exp_shift.py runs (GPU box)
cd "$(dirname "$0")/.."
( for i in $(seq 1 60); do /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|memory)" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done ) > gpurun_out/clock_samples.txt &
SP=$!
for i in 1 2 3 4; do timeout 100 python tools/exp_shift.py 2>&1 | tail -1; done
kill $SP 2>/dev/null
sort gpurun_out/clock_samples.txt | uniq -c | sort -rn | head -12
