"""Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_regs.py adapter4rec_amd/csrc/a4r_adapter_fused.hip [substring]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-c', src, '-o', '/dev/null',
                      '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
cur, d = None, {}
for l in out.splitlines():
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = m.group(1)
        d[cur] = {}
    for k, pat in (('vgpr', r'remark:\s+VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('spill', r'VGPRs Spill: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                   ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, l)
        if m and cur:
            d[cur][k] = int(m.group(1))
for k, v in d.items():
    if flt in k:
        name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip() or k
        print(f'{name[:100]:100s} {v}')
