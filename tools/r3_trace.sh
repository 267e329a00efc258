set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export RT_WF_KERNEL=2 RT_WF_POOLS=1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/wf_rate.py tri_mesh 1000000 1920 1080 4 > $O/trace.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wf_intersect" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("launches", len(d), "total ms", sum(d) / 1e3, "max us", max(d))
print("first 40 durations (us):", [round(x) for x in d[:40]])
print("grid sizes:", [r.get("Grid_Size") or r.get("Grid_Size_X") for r in rows[:40]])
PY
