"""top rows of a rocprofv3 kernel_stats.csv: python3 scratch/kstats_top.py file n_rows steps"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n, steps = int(sys.argv[2]), float(sys.argv[3])
for r in rows[:n]:
    print(f"{r['Name'][:85]:85s} calls {int(r['Calls']):5d} avg {float(r['AverageNs']) / 1e3:8.1f} us  per-step {float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms")
