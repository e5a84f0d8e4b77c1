"""Per-kernel averages of every counter in a rocprofv3 counter_collection.csv (used by scripts/gpu/visit.sh pmc steps)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
order = sorted(agg, key=lambda k: -sum(sum(v) for v in agg[k].values()))
for k in order[:24]:
    print(f"{k:70s} " + "  ".join(f"{c}: n={len(v)} avg={sum(v)/len(v):.4g}" for c, v in sorted(agg[k].items())))
