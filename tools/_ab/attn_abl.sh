for a in 0 1 2 4 8 6 14 15; do echo "ABL $a"; A4R_ATTN_LONG_ABL=$a python3 tools/attn_long_scaling.py 2>&1 | tail -1; done
