"""Summarise rocprofv3 --pmc CSV output: per (run, counter, kernel) launches, mean / min / max in KiB.
usage: pmc_summary.py <dir with */*counter_collection.csv, run name = first directory component> [build id]
The build id (gfdm_hip_build_id() of the library the counters were taken with; default: the library in this tree) goes into every row:
bench.py quotes a row only when it equals the id of the library it has loaded."""
import csv, glob, os, re, sys
from collections import defaultdict
root = sys.argv[1]
if len(sys.argv) > 2:
    build_id = sys.argv[2]
else:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gr-gfdm_amd", "python"))
    import gfdm_amd
    build_id = gfdm_amd.build_id()
acc = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    run = os.path.relpath(f, root).split(os.sep)[0]
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("gfdm::(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*$", "", name)
        acc[(run, r["Counter_Name"], name)].append(float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["run", "counter", "kernel", "launches", "mean_KiB", "min_KiB", "max_KiB", "build_id"])
for (run, c, k), v in sorted(acc.items()):
    if k.startswith(("k_", "copy_")):
        w.writerow([run, c, k, len(v), "%.2f" % (sum(v) / len(v)), "%.2f" % min(v), "%.2f" % max(v), build_id])
