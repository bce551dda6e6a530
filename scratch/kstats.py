"""prints `name  calls  average_us` of the kernels in a rocprofv3 kernel_stats.csv whose name matches argv[2] (regex)"""
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:8.1f} us", end="; ")
print()
