"""Sum (and average) the rows of rocprofv3 --pmc counter_collection.csv files per kernel and counter (development aid).
  python tools/pmc_summary.py DIR [kernel-name-substring ...]     default substrings: wf_ render_kernel"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
want = sys.argv[2:] or ["wf_", "render_kernel"]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        if not any(s in k for s in want):
            continue
        a = acc[(k, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("%-50s %-24s sum=%.6g launches=%d mean=%.6g" % (k, c, v, n, v / n))
