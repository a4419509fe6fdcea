"""Print per-kernel averages of every counter in a rocprofv3 --pmc counter_collection.csv (diagnostic).
usage: pmc_kernel.py <csv> [kernel substring]"""
import collections, csv, sys
sub = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if sub in r["Kernel_Name"]:
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in d.items():
    print(k[:80])
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} avg={sum(v) / len(v):.4g}")
