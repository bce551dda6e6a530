"""Per-kernel means of rocprofv3 --pmc counters from a counter_collection CSV -> JSON (stdout)."""
import csv, json, sys, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in rows.items():
    n = max(len(v) for v in cs.values())
    if not re.search(sys.argv[2] if len(sys.argv) > 2 else ".", k):
        continue
    out[k[:120]] = {"dispatches": n, **{c: sum(v) / len(v) for c, v in cs.items()}}
print(json.dumps(out, indent=1))
