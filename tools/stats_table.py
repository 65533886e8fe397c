"""Print a rocprofv3 kernel_stats.csv as a table of per-step times (usage: stats_table.py file.csv [steps=0: the optimizer kernel's launch count])."""
import csv, subprocess, sys
rows = list(csv.DictReader(open(sys.argv[1])))
# steps of the run = launches of the optimizer kernel (one per step: warm-up + timed + bench.py's instrumented and host-enqueue-burst steps), unless given
adam = [int(r['Calls']) for r in rows if 'adam' in r['Name'] and 'kernel' in r['Name']]
steps = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else (max(adam) if adam else 20)
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total GPU time per step {tot / steps / 1e6:.3f} ms')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    n = r['Name']
    if n.startswith('_Z'):
        n = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip() or n
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{n[:70]:70s} calls/step {int(r['Calls']) / steps:7.1f} avg_us {float(r['AverageNs']) / 1e3:8.1f} ms/step {float(r['TotalDurationNs']) / steps / 1e6:7.3f} {float(r['TotalDurationNs']) / tot * 100:5.1f}%")
