#!/bin/bash
# what the adapter weight gradients cost the step: side stream (A4R_WGRAD_STREAM=1) / single stream (default) / skipped (wrong gradients: timing only)
A4R_WGRAD_STREAM=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side stream  ', d['ms_per_step'])"
A4R_WGRAD_STREAM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single stream', d['ms_per_step'])"
A4R_DEBUG_SKIP_WGRAD=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skipped      ', d['ms_per_step'])"
