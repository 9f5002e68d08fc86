#!/bin/bash
# GPU box: socket power, clocks and temperatures sampled while the training step runs (1,500 steps of bench.py in the background).
#   bash tools/power_clock_samples.sh   ->  gpurun_out/r06_power_clock_samples.txt
rocm-smi --showmaxpower 2>/dev/null | grep -E 'Max' > gpurun_out/r06_power_clock_samples.txt
python3 bench.py --steps 1500 --warmup 10 --no-extras --no-pmc --no-cpu-baseline --no-profile > gpurun_out/pw_bench.json 2>/dev/null &
BP=$!
sleep 16
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|memory)" ; echo ---; sleep 2; done >> gpurun_out/r06_power_clock_samples.txt
wait $BP
tail -c 200 gpurun_out/pw_bench.json
