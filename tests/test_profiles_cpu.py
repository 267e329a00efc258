"""The committed counter summaries that bench.py folds into its roofline object (profiles/pmc_summary.json, written by
tools/make_profiles.py from the rocprofv3 --pmc passes of tools/prof_round.sh): every workload bench.py prints a line
for has an entry, every entry has the fields bench.py reads, and the figures are in range.  Whether the passes describe
the sources in the tree is reported (bench.py prints it as roofline.pmc_matches_sources), as a warning, not a failure:
a source edit after the last GPU call cannot re-take them."""
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "profiles", "pmc_summary.json")
BENCH_LINES = ("cornell", "cornell_smoke", "tri_mesh", "final")


def _table():
    with open(TABLE) as f:
        return json.load(f)


def _entry(table, workload):
    hit = [v for k, v in table.items() if k.startswith(workload + "_") and k[len(workload) + 1][0].isdigit()]
    return hit[0] if hit else next((v for v in table.values() if v.get("stands_for") == workload), None)


def test_every_bench_line_has_counters():
    table = _table()
    for wl in BENCH_LINES:
        e = _entry(table, wl)
        assert e is not None, wl
        for key in ("kernel", "rays", "traffic_bytes_per_ray", "valu_busy", "lane_efficiency", "lane_instructions_per_ray",
                    "sources_sha16", "commit", "counted_on", "source"):
            assert key in e, (wl, key)
        assert e["rays"] > 0 and e["traffic_bytes_per_ray"] > 0 and e["lane_instructions_per_ray"] > 0
        # 4 x SQ_ACTIVE_INST_VALU over the SIMD cycles is not calibrated past 1 (instructions that hold a pipe for other than
        # four cycles): tools/make_profiles.py labels those "saturated" in valu_issue, and the figure stays a little above 1
        # (profiles/r03_valu_costs.jsonl: 2.4 to 9.4 cycles an instruction; 1.19 on the Cornell kernel at 7 waves per SIMD)
        assert 0.0 < e["valu_busy"] <= 1.3 and 0.0 < e["lane_efficiency"] <= 1.0
        assert e["valu_issue"] == "saturated" or e["valu_busy"] < 0.98
        # the counter sums the entry was derived from are committed beside the table
        short = {"cornell": "cornell", "cornell_smoke": "smoke", "tri_mesh": "tri", "final": "final"}[wl]
        assert os.path.exists(os.path.join(ROOT, "profiles", "r03_%s_pmc.txt" % short)), wl


def test_wavefront_entries_carry_the_row_counts():
    table = _table()
    for wl in ("tri_mesh", "final"):
        e = _entry(table, wl)
        assert e["kernel"] == "wf_path_kernel"
        rows = e["rows_per_ray"]
        assert rows["node4_128B"] > 0 and rows["leaf_64B"] > 0 and "state_64B" in rows
        assert 0.0 <= e["l2_hit_rate"] <= 1.0


def test_counter_passes_describe_these_sources():
    sys.path.insert(0, ROOT)
    try:
        import bench
    finally:
        sys.path.remove(ROOT)
    now = bench.sources_sha16()
    stale = sorted(k for k, v in _table().items() if v.get("sources_sha16") != now)
    if stale:
        warnings.warn("profiles/pmc_summary.json was counted on other sources than cuda-raytracer_amd/csrc holds now (%s): "
                      "re-run tools/prof_round.sh + tools/make_profiles.py" % ", ".join(stale))
