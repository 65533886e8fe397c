#!/bin/bash
# socket power / clocks while the benchmark step runs: bash tools/power_sample.sh [bench.py args]     (rocm-smi readings every 0.5 s)
python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-roofline "$@" > /tmp/power_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $BP
tail -1 /tmp/power_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], 'ms/step', d['value'])"
