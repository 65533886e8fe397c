#!/bin/bash
# on the GPU box: every tools/_ab/liba4r_w4_*.so through tools/w4/w4_check.py (quick: two plain shapes), one line per library and shape
M=${M:-40448}
for so in ${LIBS:-tools/_ab/liba4r_w4_*.so}; do
  echo "== $so"
  A4R_LIB_PATH=$so timeout 120 python tools/w4/w4_check.py $M ${ROUNDS:-2} quick 2>&1 | grep -v amdgpu.ids
done
