#!/usr/bin/env python3
"""Condense rocprofv3 outputs merged under gpurun_out/ into the small files kept in profiles/.

usage: tools/summarize_profiles.py TAG STATS_DIR FETCH_DIR WRITE_DIR SQ_DIR BENCH_JSON
"""
import collections
import csv
import glob
import json
import re
import shutil
import sys


def short(k):
    # (template arguments kept: k_bucket_build<false> / <true> and k_ingest<true> / <false> are different launches)
    m = re.search(r"(k_[a-z0-9_]+(?:<[^>(]*>)?|__amd_rocclr_[A-Za-z]+)", k)
    return m.group(1).replace(" ", "") if m else k[:40]


def counters(d):
    f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(short(r["Kernel_Name"]), r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for (k, _d, c), v in per.items():
        agg[k][c].append(v)
    out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
    for k, cs in agg.items():   # dispatches of the kernel in the profiled run (every counter saw each of them once)
        out[k]["_dispatches"] = max(len(v) for v in cs.values())
        # ... and the average over the launches that did their work (see full_launches): counter above 30 % of its maximum
        for c, v in cs.items():
            full = [x for x in v if x > 0.3 * max(v)] or v
            out[k]["_full_" + c] = sum(full) / len(full)
            out[k]["_full_n_" + c] = len(full)
    return out


def durations(d):
    f = glob.glob(f"{d}/*/*kernel_trace.csv")[0]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in dur.items()}


def full_launches(d):
    """rocprofv3's kernel_stats averages over ALL launches of a kernel.  With the rotating headline a build whose
    hinted geometry is rejected launches the partition / bucket kernels once more than it needs them: those launches
    return at once (a few us) and pull the averages down.  Per kernel: the median of the launches that did their work
    (longer than 30 % of the kernel's longest) and how many of each kind there were."""
    f = glob.glob(f"{d}/*/*kernel_trace.csv")[0]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    out = {}
    for k, v in dur.items():
        top = max(v)
        full = sorted(x for x in v if x > 0.3 * top)
        out[k] = {"launches": len(v), "full_launches": len(full), "returned_at_once": len(v) - len(full),
                  "median_us_of_full_launches": full[len(full) // 2], "average_us_all_launches": sum(v) / len(v)}
    return out


def main():
    tag, stats_dir, fetch_dir, write_dir, sq_dir, bench_json = sys.argv[1:7]
    what = sys.argv[7] if len(sys.argv) > 7 else "bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline: the headline scene only, 10 M planar points"
    shutil.copy(glob.glob(f"{stats_dir}/*/*kernel_stats.csv")[0], f"profiles/{tag}_kernel_stats.csv")
    shutil.copy(bench_json, f"profiles/{tag}_bench_under_rocprof.json")
    json.dump({"_note": "kernel_stats.csv averages over all launches; launches that return at once (the doomed kernels "
                        "of a build whose hinted geometry was rejected, side-stream instances over empty lists) are "
                        "separated here: median of the launches that did their work",
               "kernels": full_launches(stats_dir)}, open(f"profiles/{tag}_kernel_full_launches.json", "w"), indent=1)
    fetch, write = counters(fetch_dir), counters(write_dir)
    out = {
        "_note": "HBM traffic per launch: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate "
                 "passes (" + what + "); per-launch averages, `dispatches` = launches of the kernel in the whole profiled run (warm-up, timed and instrumented steps); counter unit KiB; gfx950 "
                 "correction of MI355X_MICROARCH.md applied to the read side (FETCH_SIZE x 2 for wide "
                 "coalesced reads; the raw value is kept), WRITE_SIZE as is.",
        "kernels": {},
    }
    for k in fetch:
        f = fetch[k].get("FETCH_SIZE", 0.0)
        w = write.get(k, {}).get("WRITE_SIZE", 0.0)
        out["kernels"][k] = {"fetch_KiB_raw": f, "fetch_bytes_corrected": 2048.0 * f, "write_bytes": 1024.0 * w,
                             "dispatches": fetch[k].get("_dispatches", 0),
                             # (launches that returned at once left out: see rNN_kernel_full_launches.json)
                             "full_launches": fetch[k].get("_full_n_FETCH_SIZE", 0),
                             "fetch_bytes_corrected_full_launch": 2048.0 * fetch[k].get("_full_FETCH_SIZE", 0.0),
                             "write_bytes_full_launch": 1024.0 * write.get(k, {}).get("_full_WRITE_SIZE", 0.0)}
    json.dump(out, open(f"profiles/{tag}_hbm_traffic.json", "w"), indent=1)
    if sq_dir == "-":   # (secondary workloads: kernel stats + traffic only)
        for k in sorted(out["kernels"], key=lambda k: -out["kernels"][k]["fetch_bytes_corrected"] * out["kernels"][k]["dispatches"])[:14]:
            print(k, json.dumps(out["kernels"][k]))
        return
    sq, dur = counters(sq_dir), full_launches(sq_dir)
    rows = {}
    for k, c in sq.items():
        if "SQ_INSTS_VALU" not in c or k not in dur:
            continue
        # (per launch that did its work: launches that returned at once are left out, see full_launches)
        c = {n[len("_full_"):]: v for n, v in c.items() if n.startswith("_full_") and not n.startswith("_full_n_")}
        ns = dur[k]["median_us_of_full_launches"] * 1e3
        cyc_per_xcd = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        clock_ghz = cyc_per_xcd / ns if ns else 0.0
        busy = (c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0) / (1024.0 * cyc_per_xcd) if cyc_per_xcd else 0.0
        rows[k] = {
            "duration_us": ns / 1e3,
            "wave_valu_instructions": c.get("SQ_INSTS_VALU"),
            "valu_busy_fraction": busy,
            "effective_clock_GHz": clock_ghz,
            "waves_per_simd": (c.get("SQ_WAVE_CYCLES", 0.0) * 4.0) / (1024.0 * cyc_per_xcd) if cyc_per_xcd else 0.0,
            "wait_any_fraction_of_wave_cycles": c.get("SQ_WAIT_ANY", 0.0) / c.get("SQ_WAVE_CYCLES", 1.0),
        }
    json.dump({"_note": "SQ counters per launch that did its work (sum over XCDs); SQ_* cycle counters are quad-cycles; "
                        "valu_busy = SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * GRBM_GUI_ACTIVE/8)",
               "kernels": rows}, open(f"profiles/{tag}_sq_counters.json", "w"), indent=1)
    # instruction mix of the VALU (a fourth --pmc pass, optional 8th argument): wave-level instructions per launch by
    # type - what the RANSAC kernel EXECUTES, to hold against the algorithmic flop count of SURVEY 8(d)
    if len(sys.argv) > 8 and sys.argv[8] != "-":
        mix_c, mix_d = counters(sys.argv[8]), full_launches(sys.argv[8])
        mix = {}
        for k, c in mix_c.items():
            if not any(n.startswith("SQ_INSTS_VALU_") for n in c) or k not in mix_d:
                continue
            mix[k] = {"duration_us": mix_d[k]["median_us_of_full_launches"], "dispatches": c.get("_dispatches", 0)}
            mix[k].update({n[len("_full_SQ_INSTS_VALU_"):].lower(): v for n, v in c.items()
                           if n.startswith("_full_SQ_INSTS_VALU_")})
        json.dump({"_note": "wave-level VALU instructions per launch by type (rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 "
                            "MUL_F64 FMA_F64 TRANS_F64 ADD_F32 MUL_F32 FMA_F32 INT32, one pass; sum over XCDs); one "
                            "wave instruction = 64 lane operations, an FMA = 2 flops per lane",
                   "kernels": mix}, open(f"profiles/{tag}_valu_mix.json", "w"), indent=1)
        for k in sorted(mix):
            if k.startswith("k_ransac<"):
                print(k, json.dumps(mix[k]))
    for k in sorted(set(rows) | set(out["kernels"])):
        if not re.match(r"k_(ransac<|ingest|part_|bucket_|compact_tiles|transpose)", k):
            continue
        if k in rows:
            print(k, json.dumps(rows[k]))
        if k in out["kernels"]:
            print(k, json.dumps(out["kernels"][k]))


if __name__ == "__main__":
    main()
