#!/bin/bash
# GPU box: where a --real-shaped step's time goes at 32 users per step (launch-boundary histogram + host profile)     bash tools/_ab/real_shaped_gaps.sh <tag>
TAG=${1:-r06_i}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
out=gpurun_out/${TAG}_real_shaped_gaps.txt
: > $out
python bench.py --real-shaped --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench.py --real-shaped:', d['ms_per_step'], 'ms/step', d['value'], 'user-seq/s')" >> $out
rm -rf gpurun_out/_trace_rs
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_trace_rs -o t -- python3 bench.py --real-shaped --steps 20 --warmup 10 --no-cpu-baseline --no-roofline > /dev/null 2>&1
f=$(find gpurun_out/_trace_rs -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f 5 >> $out 2>&1
rm -rf gpurun_out/_trace_rs
python -m cProfile -o /tmp/rs.prof bench.py --real-shaped --steps 300 --warmup 20 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python -c "
import pstats; st=pstats.Stats('/tmp/rs.prof'); st.sort_stats('tottime').print_stats(25)" 2>&1 | grep -v "^$" | head -45 >> $out
cat $out
