"""Split a rocprofv3 kernel_trace.csv by kernel, launch size (grid) and queue: one bench run launches every kernel in several
situations (4096 blocks alone on the GPU, inside the multi-stream headline loop, 65 536 blocks), which the per-name
averages of rocprofv3's own stats file mix.  usage: trace_by_shape.py <kernel_trace.csv> > by_launch_shape.csv"""
import csv, re, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
col = lambda r, *names: next(r[n] for n in names if n in r)
groups = defaultdict(list)
for r in rows:
    name = col(r, "Kernel_Name")
    name = name.replace("gfdm::(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)                      # drop the argument list
    grid = int(col(r, "Grid_Size_X", "Grid_Size"))
    wg = int(col(r, "Workgroup_Size_X", "Workgroup_Size"))
    queue = col(r, "Queue_Id")
    dur = (int(col(r, "End_Timestamp")) - int(col(r, "Start_Timestamp"))) / 1e3
    groups[(name, grid // wg, wg, queue)].append(dur)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "workgroups", "workgroup_size", "queue", "launches", "mean_us", "median_us", "min_us", "max_us"])
for (name, g, wg, q), d in sorted(groups.items(), key=lambda kv: (kv[0][0], kv[0][1], kv[0][3])):
    d.sort()
    w.writerow([name, g, wg, q, len(d), "%.2f" % (sum(d) / len(d)), "%.2f" % d[len(d) // 2], "%.2f" % d[0], "%.2f" % d[-1]])
