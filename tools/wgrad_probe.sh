#!/bin/bash
# what the adapter weight gradients cost the step: side stream (A4R_WGRAD_STREAM=1) / single stream (default).  (The third, wrong-gradient 'skipped' leg of round 4 was a library env knob; it is gone from the library -- profiles/LOG.md keeps its numbers.)
A4R_WGRAD_STREAM=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side stream  ', d['ms_per_step'])"
A4R_WGRAD_STREAM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single stream', d['ms_per_step'])"
