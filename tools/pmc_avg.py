"""Average rocprofv3 --pmc counters per kernel: pmc_avg.py <counter_collection.csv> [kernel substring]"""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
        continue
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, "launches", max(len(v) for v in d.values()))
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v) / len(v):16.1f}")
