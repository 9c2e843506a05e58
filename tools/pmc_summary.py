"""Mean counter value per dispatch, per kernel, from rocprofv3 --pmc CSV output
(files *_counter_collection.csv under the given directories)."""
import csv, glob, sys
from collections import defaultdict

acc = defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
            acc[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:46s} {c:22s} n={len(v):5d} mean={sum(v) / len(v):.1f}")
