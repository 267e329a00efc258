"""Usage: python tools/make_profiles.py TAG [node4_rows_per_ray leaf_rows_per_ray]   (the two numbers from tools/stamps.py on the triangle scene)
Copy the rocprofv3 summaries of a tools/prof_round.sh run from gpurun_out/<tag>/ into profiles/ (tracked):
kernel-trace stats, one text file of PMC sums per kernel, and profiles/hbm_traffic.json (bytes per render call,
read by bench.py as roofline.traffic).  FETCH_SIZE / WRITE_SIZE are in KiB.  Corrections (MI355X_MICROARCH.md,
HBM): the Cornell megakernel reads its accumulation buffer as a coalesced stream, for which gfx950 tallies half
the bytes -> FETCH doubled; the wavefront kernels gather 64-byte rows, and there FETCH_SIZE agrees with
TCC_MISS x 64 B (656 GB vs 618 GB).  Calibrated in round 2 (tools/ubench/gather.hip, profiles/r02_gather_pmc.txt): FETCH_SIZE
= 64 B x (requests that miss the L2), whether the request is a 64-B row or a 128-B row, and whether it is then served by the
Infinity Cache or by HBM.  So for the wavefront kernels the bytes that really cross the fabric are FETCH_SIZE x
(128 N + 64 L) / (64 (N + L)) with N four-wide node rows and L leaf rows per ray (tools/stamps.py).  Every entry is stamped
with the commit the passes were taken at."""
import collections, csv, glob, json, os, shutil, sys
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rows_node4 = float(sys.argv[2]) if len(sys.argv) > 3 else 17.75      # measured (1M triangles, 1920x1080): tools/stamps.py
rows_leaf = float(sys.argv[3]) if len(sys.argv) > 3 else 7.36
try:
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    if subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "cuda-raytracer_amd"], text=True).strip():
        commit += "+uncommitted"
except Exception:
    commit = "unknown"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

def sums(d):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if not any(s in k for s in ("wf_", "render_kernel")):
                continue
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc

for name in ("trace_default", "trace_tri"):
    for f in glob.glob(os.path.join(src, name, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.reader(open(f)))
        keep = [rows[0]] + [r for r in rows[1:] if any(s in r[0] for s in ("wf_", "render_kernel", "resolve_kernel"))]
        with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)), "w", newline="") as o:
            csv.writer(o, quoting=csv.QUOTE_ALL).writerows(keep)

allsum = {}
for wl, dirs in (("cornell", ["cornell_fetch", "cornell_write", "cornell_sq", "cornell_sq2"]),
                 ("tri", ["tri_fetch", "tri_write", "tri_tcc", "tri_sq", "tri_grbm"])):
    with open(os.path.join(dst, "%s_%s_pmc.txt" % (tag, wl)), "w") as o:
        o.write("# rocprofv3 --pmc sums over all launches of one bench.py render call (tools/prof_round.sh); one pass per counter group\n")
        for d in dirs:
            a = sums(d)
            for (k, c), (v, n) in sorted(a.items()):
                o.write("%-34s %-24s sum=%.6g launches=%d\n" % (k[:34], c, v, n))
                allsum[(wl, k.split("<")[0].replace("void ", ""), c)] = v
def g(wl, k, c):
    return allsum.get((wl, k, c), 0.0)

def derived(wl, k):
    """VALU busy share of the SIMDs (SQ_ACTIVE_INST_VALU counts quad-cycles; GRBM_GUI_ACTIVE is summed over the
    8 XCDs), lane efficiency (active lanes per VALU instruction / 64) and, where counted, the L2 hit rate"""
    simd_cycles = g(wl, k, "GRBM_GUI_ACTIVE") / 8.0 * 1024.0
    busy = 4.0 * g(wl, k, "SQ_ACTIVE_INST_VALU") / simd_cycles if simd_cycles else None
    out = {"valu_busy": busy,
           # the formula is not calibrated beyond 1 (packed and transcendental instructions hold a pipe for other than 4
           # cycles): at or above 1 the only thing it says is "an instruction issues in every cycle"
           "valu_issue": None if busy is None else ("saturated" if busy >= 0.98 else "%.0f %% busy" % (100.0 * busy)),
           "lane_efficiency": g(wl, k, "SQ_THREAD_CYCLES_VALU") / (64.0 * g(wl, k, "SQ_ACTIVE_INST_VALU")) if g(wl, k, "SQ_ACTIVE_INST_VALU") else None,
           "wait_share_of_wave_cycles": g(wl, k, "SQ_WAIT_ANY") / g(wl, k, "SQ_WAVE_CYCLES") if g(wl, k, "SQ_WAVE_CYCLES") else None}
    if g(wl, k, "TCC_REQ_sum"):
        out["l2_hit_rate"] = g(wl, k, "TCC_HIT_sum") / g(wl, k, "TCC_REQ_sum")
    return out

gather_fix = (128.0 * rows_node4 + 64.0 * rows_leaf) / (64.0 * (rows_node4 + rows_leaf))
tri_fetch = sum(v for (wl, k, c), v in allsum.items() if wl == "tri" and c == "FETCH_SIZE") * 1024.0
tri_fetch_isect = g("tri", "wf_intersect_kernel", "FETCH_SIZE") * 1024.0
tri_write = sum(v for (wl, k, c), v in allsum.items() if wl == "tri" and c == "WRITE_SIZE") * 1024.0
summary = {
    "cornell_1024x1024_1024spp": dict(derived("cornell", "render_kernel"), kernel="render_kernel", source=tag, commit=commit,
        traffic_bytes=(2.0 * g("cornell", "render_kernel", "FETCH_SIZE") + g("cornell", "render_kernel", "WRITE_SIZE")) * 1024.0),
    "tri_mesh_1920x1080_256spp": dict(derived("tri", "wf_intersect_kernel"), kernel="wf_intersect_kernel", source=tag, commit=commit,
        # path state in the intersect kernel: ray_o + ray_d read by the refill, hit written: three 16-B pieces per ray, and the
        # queue stays sorted by slot, so four neighbouring rays share each 64-B line: 3 / 4 requests per ray
        rows_per_ray={"node4_128B": rows_node4, "leaf_64B": rows_leaf, "state_64B": 0.75, "source": "tools/stamps.py (diagnostic build)"},
        traffic_uncorrected_bytes=tri_fetch + tri_write,
        traffic_bytes=tri_fetch - tri_fetch_isect + tri_fetch_isect * gather_fix + tri_write,
        fetch_correction_intersect=gather_fix),
}
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
