"""Counter values per dispatch (not their mean) of the kernels whose name contains a string,
from rocprofv3 --pmc CSV output:  python tools/pmc_rows.py DIR NAME_PART [FIRST_N]"""
import csv, glob, sys
from collections import defaultdict

rows = defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = int(float(r["Counter_Value"]))
            rows[int(r["Dispatch_Id"])]["grid"] = r.get("Grid_Size")
for d in sorted(rows)[:int(sys.argv[3]) if len(sys.argv) > 3 else None]:
    print(d, rows[d])
