#!/usr/bin/env python
"""Where a LAUNCH of the fused adapter forward spends its time outside the tile loop (diagnostic build -DA4R_STAMP=3): wave 0 of every workgroup
stamps s_memrealtime at entry, after the prologue, after its first tile's first barrier, at the end of its first tile and after its last tile.
usage: A4R_LIB_PATH=tools/_ab/liba4r_adstamp3.so python tools/adapter_launch_timeline.py [M]
       A4R_LIB_PATH=tools/_ab/liba4r_adstamp4.so python tools/adapter_launch_timeline.py [M] bwd       (-DA4R_STAMP=4: the backward launch)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
BWD = sys.argv[-1] == 'bwd'
if BWD:
    sys.argv = sys.argv[:-1]
import numpy as np
import adapter_bench as AB
from adapter4rec_amd import _lib as L
import torch

us = AB.timeit(AB.bwd_fused_y if BWD else AB.fwd_fused_y)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (256 * 2 * 8))()
assert L.lib().a4r_debug_adapter_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16).astype(np.int64) / 100.0
t0 = st[:, 0].min()
q = lambda a: f'median {np.median(a):6.2f}  min {a.min():6.2f}  max {a.max():6.2f}'
print(f'launch period (events, back to back) {us:.1f} us;  first entry -> last exit {st[:, 2].max() - t0:.2f} us')
print('entry after the first workgroup     ', q(st[:, 0] - t0))
if BWD:
    print('entry -> every prologue request issued', q(st[:, 5] - st[:, 0]))
    print('prologue (entry -> weight image + parameters in LDS)', q(st[:, 1] - st[:, 0]))
    print('-> first tile done                   ', q(st[:, 4] - st[:, 1]))
    print('-> last tile done                    ', q(st[:, 2] - st[:, 4]))
    print('exit before the last workgroup       ', q(st[:, 2].max() - st[:, 2]))
    sys.exit(0)
print('last wave: entry after wave 0         ', q(st[:, 8] - st[:, 0]))
print('last wave: entry -> requests issued   ', q(st[:, 13] - st[:, 8]))
print('last wave: -> its parameters arrived  ', q(st[:, 14] - st[:, 13]))
print('entry -> every prologue request issued', q(st[:, 5] - st[:, 0]))
print('-> first loads (parameters) arrived   ', q(st[:, 6] - st[:, 5]))
print('prologue (entry -> parameters in LDS)', q(st[:, 1] - st[:, 0]))
print('-> first tile past barrier 1         ', q(st[:, 3] - st[:, 1]))
print('-> first tile done                   ', q(st[:, 4] - st[:, 3]))
print('-> last tile done                    ', q(st[:, 2] - st[:, 4]))
print('exit before the last workgroup       ', q(st[:, 2].max() - st[:, 2]))
