ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/pa -o a --output-format csv -- python3 $ROOT/tools/attn_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC --kernel-trace -d $OUT/pb -o b --output-format csv -- python3 $ROOT/tools/attn_bench.py > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/pc -o c --output-format csv -- python3 $ROOT/tools/attn_bench.py > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -d $OUT/pd -o d --output-format csv -- python3 $ROOT/tools/attn_bench.py > /dev/null 2>&1
python3 $ROOT/tools/pmc_sq.py attn_long $OUT/pa $OUT/pb $OUT/pc $OUT/pd
rm -rf $OUT/pa $OUT/pb $OUT/pc $OUT/pd
