#!/bin/bash
# rocprofv3 kernel statistics (+ optional PMC traffic passes) of the bench step, on the GPU box:
#   gpurun -- 'bash tools/profile_step.sh r02_b [pmc|nopmc] [workload] [dtype]'
# writes gpurun_out/<tag>_kernel_stats.csv, <tag>_bench.json (+ <tag>_pmc_hbm_traffic.json); copy what is to be judged to profiles/.
set -u
TAG=${1:-prof}; PMC=${2:-}; WL=${3:-bert_houlsby}; DT=${4:-bf16}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o s --output-format csv -- python3 $ROOT/bench.py --steps 15 --warmup 3 --no-cpu-baseline --workload $WL --dtype $DT > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cp $(find $OUT/${TAG}_prof -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
if [ "$PMC" = "pmc" ]; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_pmc_f -o f --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload $WL > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_pmc_w -o w --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload $WL > /dev/null 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_f $OUT/${TAG}_pmc_w > $OUT/${TAG}_pmc_hbm_traffic.json
  rm -rf $OUT/${TAG}_pmc_f $OUT/${TAG}_pmc_w
  # matrix-pipe utilisation (north_star: "rocprof reporting ... MFMA utilisation"): one counter per pass, program directly after --
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $OUT/${TAG}_pmc_m -o m --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload $WL > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${TAG}_pmc_g -o g --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --workload $WL > /dev/null 2>&1
  python3 $ROOT/tools/pmc_mfma.py $OUT/${TAG}_pmc_m $OUT/${TAG}_pmc_g > $OUT/${TAG}_pmc_mfma_util.json
  rm -rf $OUT/${TAG}_pmc_m $OUT/${TAG}_pmc_g
fi
rm -rf $OUT/${TAG}_prof
cat $OUT/${TAG}_bench.json
