#!/bin/bash
# GPU box: CPU enqueue time per step and the launch-boundary histogram of the headline step at 32 and 8 users   bash tools/_ab/launch_gaps.sh <tag>
TAG=${1:-r06_c}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
: > gpurun_out/${TAG}_launch_gaps.txt
for B in 32 8; do
  python tools/cpu_enqueue.py $B 0 2>/dev/null | tail -1 >> gpurun_out/${TAG}_launch_gaps.txt
  rm -rf gpurun_out/_trace_$B
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_trace_$B -o t -- python3 bench.py --batch $B --steps 20 --warmup 10 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  f=$(find gpurun_out/_trace_$B -name "*kernel_trace.csv" | head -1)
  echo "---- users per step: $B ($f)" >> gpurun_out/${TAG}_launch_gaps.txt
  python tools/trace_gaps.py $f 5 >> gpurun_out/${TAG}_launch_gaps.txt 2>&1
  rm -rf gpurun_out/_trace_$B
done
cat gpurun_out/${TAG}_launch_gaps.txt
