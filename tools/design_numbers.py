#!/usr/bin/env python3
"""The numbers DESIGN.md quotes, out of the tracked profile files of a round (profiles/rNN_*): the per-kernel table of
one step of the headline and the one-liners of the secondaries.   usage: tools/design_numbers.py r06"""
import csv, json, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
import os
# (round 6: the bounded line is TAG_bench.json, the full result TAG_bench_detail.json)
b = json.load(open(f"profiles/{tag}_bench_detail.json" if os.path.exists(f"profiles/{tag}_bench_detail.json")
               else f"profiles/{tag}_bench.json"))
tr = json.load(open(f"profiles/{tag}_hbm_traffic.json"))["kernels"]
sq = json.load(open(f"profiles/{tag}_sq_counters.json"))["kernels"]
st = {}
for r in csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv")):
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(" ", "")
    st[name] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
try:
    fl = json.load(open(f"profiles/{tag}_kernel_full_launches.json"))["kernels"]
except OSError:
    fl = {}
print("| kernel | µs (median of its full launches) | fetch / write MB | counter TB/s (of 8) | waves/SIMD, wait | bound by |")
print("|---|---|---|---|---|---|")
for k in ("k_build_begin", "k_part_hist<true,true>", "k_part_hist<false,true>", "k_table_scan", "k_part_scatter<8,false,12,false>",
          "k_bucket_build", "k_bucket_scan_totals", "k_bucket_finish<false>", "k_block_prepare", "k_block_scatter",
          "k_ransac<64,16,6,true,true>", "k_ransac<128,8,6,true,true>", "k_blk_kept", "k_scan_lookback",
          "k_compact_tiles", "k_blk_compact"):
    us = st.get(k, (None, 0))
    if k in fl:
        us = (fl[k]["median_us_of_full_launches"], fl[k]["full_launches"])
    t = tr.get(k)
    q = sq.get(k)
    fb = t.get("fetch_bytes_corrected_full_launch", t["fetch_bytes_corrected"]) if t else 0
    wb = t.get("write_bytes_full_launch", t["write_bytes"]) if t else 0
    mb = f"{fb / 1e6:.0f} / {wb / 1e6:.0f}" if t else "—"
    tbs = f"{(fb + wb) / (us[0] * 1e-6) / 1e12:.2f}" if t and us[0] else "—"
    occ = f"{q['waves_per_simd']:.2f}, {100 * q['wait_any_fraction_of_wave_cycles']:.0f} %" if q else "—"
    print(f"| `{k}` | {us[0]:.1f} ({us[1]} launches) | {mb} | {tbs} | {occ} | |" if us[0] else f"| `{k}` | — |")
print()
print("headline: %.3f ms per step, %.0f Mpoints/s, launches %.1f, host waits %.2f" % (
    b["ms_per_step"], b["value"], b.get("launches_per_step") or -1, b["host_syncs_per_step"]))
print("roofline: %.1f GB/s = %.4f; launch %.3f ms; traffic %s" % (b["roofline"]["achieved"], b["roofline"]["frac"],
                                                                   b["roofline"]["launch_ms"], b["roofline"]["traffic"]))
s = b["secondary"]
for k in ("same_cloud", "insert_subdivide_only", "insert_subdivide_general_path", "pcie_inclusive", "pcie_pipelined",
          "api_inclusive", "api_pipelined", "api_pipelined_2ctx", "uniform_scene", "sparse_scene", "two_streams"):
    v = s.get(k)
    if v:
        print(k, {a: (round(x, 3) if isinstance(x, float) else x) for a, x in v.items() if a != "note"})
nh = s["no_geometry_hint"]
print("no hint:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in nh.items() if k not in ("legs", "note")})
c1 = s["c1_octree_100k"]
print("c1:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in c1.items() if k not in ("note", "kernels_ms")})
print("c4 subdivide ms", round(s["c4_manager"]["subdivide_ms"], 3), "| c5 shard step / build ms", round(s["c5_shard"]["ms"], 2),
      round(s["c5_shard"]["insert_subdivide_only_ms"], 2))
rb = b["roofline_build"]
print("build:", rb["whole_build"])
print({k: (round(v["ms_per_step"], 4), v["counter_bytes_per_point"] and round(v["counter_bytes_per_point"], 1),
           v["counter_frac_of_8TBs"] and round(v["counter_frac_of_8TBs"], 3)) for k, v in rb["all"].items()})
print("executed:", json.dumps(b["roofline_valu"].get("executed"), indent=1))
print("cpu:", b["cpu_baseline"])
