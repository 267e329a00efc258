"""Usage: python tools/make_profiles.py TAG [OUT_PREFIX]
Copy the rocprofv3 summaries of a tools/prof_round.sh run (both parts) from gpurun_out/<TAG>/ into profiles/ (tracked) as
<OUT_PREFIX>_* (default: TAG): kernel-trace stats, one text file of PMC sums per kernel and workload, and
profiles/pmc_summary.json, which bench.py reads for roofline.traffic / roofline.pmc / fabric_measured / issue_ceiling_frac.
One entry per line of the default bench.py run: cornell (headline), cornell_smoke, tri_mesh, final.  The final scene's passes
are taken at 1920x1080 x 64 spp (the configs[4] frame is a minute a pass); everything bench.py uses from an entry is per ray,
and the entry says which command it was counted on.

FETCH_SIZE / WRITE_SIZE are in KiB.  Corrections (MI355X_MICROARCH.md, HBM): the megakernel reads its accumulation buffer as
a coalesced stream, for which gfx950 tallies half the bytes -> FETCH doubled; the wavefront kernels gather 64-byte rows, and
there FETCH_SIZE agrees with TCC_MISS x 64 B.  Calibrated in round 2 (tools/ubench/gather.hip, profiles/r02_gather_pmc.txt):
FETCH_SIZE = 64 B x (requests that miss the L2), whether the request is a 64-B row or a 128-B row, and whether it is then
served by the Infinity Cache or by HBM.  So for the intersect kernel the bytes that really cross the fabric are FETCH_SIZE x
(128 N + 64 L) / (64 (N + L)) with N four-wide node rows and L leaf rows per ray (tools/stamps.py, run by prof_round.sh).
Every entry is stamped with the commit the passes were taken at."""
import collections, csv, glob, json, os, re, sys
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else tag
try:
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    if subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "cuda-raytracer_amd"], text=True).strip():
        commit += "+uncommitted"
except Exception:
    commit = "unknown"
def sources_sha16():
    """hash of the device and host sources the library is built from: bench.py recomputes it and says whether the counter
    passes describe the code it is timing (roofline.pmc_matches_sources)"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "cuda-raytracer_amd", "csrc", "*"))):
        if f.endswith((".h", ".hip", ".cpp")):
            h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


sources = sources_sha16()
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
KERNELS = ("wf_", "render_kernel", "resolve_kernel")


def sums(d):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if not any(s in k for s in KERNELS[:2]):
                continue
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc


def bench_line(log):
    """the JSON line bench.py printed under the profiler (rays of the counted command = value x time)"""
    try:
        for line in open(os.path.join(src, log)):
            if line.startswith("{") and '"metric"' in line:
                return json.loads(line)
    except OSError:
        pass
    return None


def rows_per_ray(name):
    try:
        m = re.search(r"rows fetched per ray: ([0-9.]+) four-wide nodes .* \+ ([0-9.]+) leaf", open(os.path.join(src, name)).read())
        return (float(m.group(1)), float(m.group(2))) if m else None
    except OSError:
        return None


for name in ("trace_default", "trace_tri", "trace_final"):
    for f in glob.glob(os.path.join(src, name, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.reader(open(f)))
        keep = [rows[0]] + [r for r in rows[1:] if any(s in r[0] for s in KERNELS)]
        with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (prefix, name)), "w", newline="") as o:
            csv.writer(o, quoting=csv.QUOTE_ALL).writerows(keep)

WORKLOADS = (("cornell", ["cornell_fetch", "cornell_write", "cornell_sq", "cornell_sq2", "cornell_grbm"]),
             ("smoke", ["smoke_fetch", "smoke_write", "smoke_sq", "smoke_grbm"]),
             ("tri", ["tri_fetch", "tri_write", "tri_tcc", "tri_sq", "tri_grbm"]),
             ("final", ["final_fetch", "final_write", "final_tcc", "final_sq", "final_grbm"]))
allsum = {}
for wl, dirs in WORKLOADS:
    lines = []
    for d in dirs:
        for (k, c), (v, n) in sorted(sums(d).items()):
            lines.append("%-34s %-24s sum=%.6g launches=%d\n" % (k[:34], c, v, n))
            allsum[(wl, k.split("<")[0].replace("void ", ""), c)] = v
    if lines:
        with open(os.path.join(dst, "%s_%s_pmc.txt" % (prefix, wl)), "w") as o:
            o.write("# rocprofv3 --pmc sums over all launches of one bench.py render call (tools/prof_round.sh); one pass per counter group\n")
            o.writelines(lines)


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


def rays_of(log):
    j = bench_line(log)
    return (j["value"] * 1e6 * j["ms_per_step"] * 1e-3 * j["steps"], j["config"]["workload"]) if j else (None, None)


summary = {}
for wl, key in (("cornell", None), ("smoke", None)):
    rays, workload = rays_of("%s_fetch.log" % wl)
    if not rays or not g(wl, "render_kernel", "FETCH_SIZE"):
        continue
    traffic = (2.0 * g(wl, "render_kernel", "FETCH_SIZE") + g(wl, "render_kernel", "WRITE_SIZE")) * 1024.0
    summary[workload] = dict(derived(wl, "render_kernel"), kernel="render_kernel", source=tag, commit=commit, sources_sha16=sources, counted_on=workload, rays=rays,
                             traffic_bytes=traffic, traffic_bytes_per_ray=traffic / rays,
                             # active lanes x VALU instructions (SQ_THREAD_CYCLES_VALU counts lanes x quad-cycles): bench.py divides by
                             # its own kernel time and by 1024 SIMDs x 16 lanes x clock -> issue_ceiling_frac
                             lane_instructions_per_ray=g(wl, "render_kernel", "SQ_THREAD_CYCLES_VALU") / rays)
for wl, stamps, stands_for in (("tri", "stamps_tri.txt", "tri_mesh"), ("final", "stamps_final.txt", "final")):
    rays, workload = rays_of("%s_fetch.log" % wl)
    rows = rows_per_ray(stamps)
    # the kernel that walks the trees: the one-kernel form's wf_path_kernel (default) or the two-kernel form's wf_intersect_kernel
    main = "wf_path_kernel" if g(wl, "wf_path_kernel", "FETCH_SIZE") else "wf_intersect_kernel"
    if not rays or not rows or not g(wl, main, "FETCH_SIZE"):
        continue
    n4, lf = rows
    gather_fix = (128.0 * n4 + 64.0 * lf) / (64.0 * (n4 + lf))
    fetch = sum(v for (w, k, c), v in allsum.items() if w == wl and c == "FETCH_SIZE") * 1024.0
    fetch_isect = g(wl, main, "FETCH_SIZE") * 1024.0
    write = sum(v for (w, k, c), v in allsum.items() if w == wl and c == "WRITE_SIZE") * 1024.0
    traffic = fetch - fetch_isect + fetch_isect * gather_fix + write
    summary[workload] = dict(derived(wl, main), kernel=main, source=tag, commit=commit, sources_sha16=sources, counted_on=workload,
        stands_for=stands_for, rays=rays,
        # path state: none in the one-kernel form (a 16-B sum per slot at its end); in the two-kernel form ray_o + ray_d read by
        # the refill and hit written, three 16-B pieces per ray with four neighbouring rays sharing each 64-B line
        rows_per_ray={"node4_128B": n4, "leaf_64B": lf, "state_64B": 0.0 if main == "wf_path_kernel" else 0.75, "source": "tools/stamps.py (diagnostic build)"},
        traffic_uncorrected_bytes=fetch + write, traffic_bytes=traffic, traffic_bytes_per_ray=traffic / rays,
        fetch_correction_intersect=gather_fix,
        logic_kernel=derived(wl, "wf_logic_kernel") if g(wl, "wf_logic_kernel", "SQ_WAVES") else None,
        valu_instructions_per_ray=g(wl, main, "SQ_INSTS_VALU") / rays,
        lane_instructions_per_ray=(g(wl, main, "SQ_THREAD_CYCLES_VALU") + g(wl, "wf_logic_kernel", "SQ_THREAD_CYCLES_VALU")) / rays)
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
