#!/bin/bash
# Timing-only ablation builds of the 256-tile GEMM's EPILOGUE forms (results are WRONG in these builds): what the GELU arithmetic,
# the second output and the Pre operand each cost.   bash tools/epi_abl.sh   (on the GPU box)
set -e
cd "$(dirname "$0")/.."
ABLS="${ABLS:-0 64 128 192 256}" BUILD_ONLY=1 bash tools/gemm_abl.sh
for abl in ${ABLS:-0 64 128 192 256}; do
  echo "ABL=$abl"
  A4R_LIB_PATH=tools/_ab/liba4r_abl$abl.so python tools/gemm_forms.py 2>&1 | grep -E "gelu|dmul" | cut -c1-100
done
