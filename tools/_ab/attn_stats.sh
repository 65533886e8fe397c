ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/ps -o s --output-format csv -- python3 $ROOT/tools/attn_bench.py > /dev/null 2>&1
python3 - <<EOP
import csv,glob
f=glob.glob("$OUT/ps/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'attn_long' in r['Name']: print('%-60s %5s %9.1f us'%(r['Name'][18:78], r['Calls'], float(r['AverageNs'])/1e3))
EOP
rm -rf $OUT/ps
