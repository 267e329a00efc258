"""Sum rocprofv3 --pmc counter_collection.csv rows per kernel name prefix (development aid)."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        if not any(s in k for s in ("wf_", "render_kernel")):
            continue
        a = acc[(k, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("%-50s %-24s sum=%.6g launches=%d" % (k, c, v, n))
