#!/usr/bin/env python
"""Where a tile of the fused adapter backward kernel spends its time (diagnostic build -DA4R_STAMP of a4r_adapter_fused.hip): waves 0 and 7 of
every workgroup stamp s_memrealtime at the top of their third tile, before / after each of its three barriers and at its end.
usage: A4R_LIB_PATH=tools/_ab/liba4r_adstamp.so python tools/adapter_timeline.py          (build -DA4R_STAMP=1: backward)
       A4R_LIB_PATH=tools/_ab/liba4r_adstamp2.so python tools/adapter_timeline.py fwd     (build -DA4R_STAMP=2: forward)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
FWD = len(sys.argv) > 1 and sys.argv[-1] == 'fwd'
sys.argv = sys.argv[:1]
import numpy as np
import adapter_bench as AB          # builds the operands and times the launches (prints its own lines)
from adapter4rec_amd import _lib as L
import torch

for _ in range(10):
    AB.fwd_fused_y() if FWD else AB.bwd_fused()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (256 * 2 * 8))()
assert L.lib().a4r_debug_adapter_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 2, 8).astype(np.int64)
if FWD:
    names = ['down-projection MFMA, partials to LDS', 'wait at barrier 1', 'reduce partials, act, zp / z', 'wait at barrier 2',
             'up-projection, residuals, row statistics', 'wait at barrier 3', 'normalise, store y']
else:
  names = ['LN part 1 (xhat, g, row sums)', 'wait at barrier 1', 'part 2: dv, store, dz MFMA, partials to LDS', 'wait at barrier 2',
         "reduce partials, act', dzp", 'wait at barrier 3', 'dh MFMA, dropout, store']
for w, nm in ((0, 'wave 0'), (1, 'wave 7')):
    d = np.diff(st[:, w, :], axis=1) / 100.0
    print(f'{nm}: tile total {np.median(st[:, w, 7] - st[:, w, 0]) / 100:.2f} us')
    for i, n in enumerate(names):
        print(f'   {n:48s} {np.median(d[:, i]):5.2f} us  [{d[:, i].min():5.2f} {d[:, i].max():5.2f}]')
