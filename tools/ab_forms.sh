#!/bin/bash
# same-box A/B of two library builds on the step's GEMM forms: tools/ab_forms.sh <other.so> [rounds]
other=${1:-tools/_ab/liba4r_old.so}; rounds=${2:-2}
for i in $(seq $rounds); do
  echo "== current"; python tools/gemm_forms.py
  echo "== $other"; A4R_LIB_PATH=$other python tools/gemm_forms.py
done
