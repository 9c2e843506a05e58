"""Prints the last `window_ms` of a rocprofv3 kernel + memory-copy trace (csv files in a
directory) as a time line: start, duration, name."""
import csv, glob, sys
d, window = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:48]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"C {r['Direction']} {int(r.get('Size', 0) or 0)} B"))
ev.sort()
end = ev[-1][1]
t0 = None
for s, e, n in ev:
    if s < end - window * 1e6:
        continue
    t0 = t0 or s
    if (e - s) > 20000 or n.startswith("C"):
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  {n}")
