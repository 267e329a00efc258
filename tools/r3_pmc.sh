# counters of the two intersect kernels side by side: bash tools/r3_pmc.sh TAG [scene args]
set -e
cd $GRAFT_REPO_ROOT
TAG=$1; shift
bash tools/r3_stamps.sh $TAG
for k in 1 2; do
  RT_WF_KERNEL=$k bash tools/ab_pmc.sh ${TAG}_k$k "$@" > gpurun_out/${TAG}_pmc_k$k.txt 2>&1 || { tail -5 gpurun_out/${TAG}_pmc_k$k.txt; exit 1; }
done
python3 - <<PY
import re
for k in (1, 2):
    v = {}
    for ln in open("gpurun_out/${TAG}_pmc_k%d.txt" % k):
        m = re.match(r"(?:void )?(wf_\w+)<.*?\s(\w+)\s+sum=([0-9.e+]+)", ln)
        if m: v[(m.group(1), m.group(2))] = float(m.group(3))
        if ln.startswith("{"): print("kernel", k, ln.strip())
    for kn in ("wf_intersect_kernel", "wf_intersect2_kernel", "wf_logic_kernel"):
        g = lambda c: v.get((kn, c), 0.0)
        if not g("SQ_ACTIVE_INST_VALU"): continue
        simd = g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0
        print("  %-22s valu_busy %.3f lane_eff %.3f wait_share %.3f valu_insts %.4g vmem_rd %.4g lds %.4g l2_hit %.3f gui_active %.4g" % (
            kn, 4.0 * g("SQ_ACTIVE_INST_VALU") / simd if simd else 0, g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")),
            g("SQ_WAIT_ANY") / max(1.0, g("SQ_WAVE_CYCLES")), g("SQ_INSTS_VALU"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_LDS"),
            g("TCC_HIT_sum") / max(1.0, g("TCC_REQ_sum")), g("GRBM_GUI_ACTIVE")))
PY
