#!/usr/bin/env python3
"""Per-kernel table out of rocprofv3 --pmc passes: tools/pmc_table.py DIR [DIR ...] (counter averages per launch)."""
import collections, csv, glob, re, sys

def short(k):
    # (template arguments kept: k_bucket_build<false> / <true> and k_ingest<true> / <false> are different launches)
    m = re.search(r"(k_[a-z0-9_]+(?:<[^>(]*>)?|__amd_rocclr_[A-Za-z]+)", k)
    return m.group(1).replace(" ", "") if m else k[:40]

tab = collections.defaultdict(dict)
dur = {}
for d in sys.argv[1:]:
    per = collections.defaultdict(float)
    for f in glob.glob(f"{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            per[(short(r["Kernel_Name"]), r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    agg = collections.defaultdict(list)
    for (k, _d, c), v in per.items():
        agg[(k, c)].append(v)
    for (k, c), v in agg.items():
        tab[k][c] = sum(v) / len(v)
    for f in glob.glob(f"{d}/*/*kernel_trace.csv"):
        dd = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            dd[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for k, v in dd.items():
            dur[k] = sum(v) / len(v) / 1e3
cols = sorted({c for k in tab for c in tab[k]})
print("kernel".ljust(28), "us".rjust(9), " ".join(c[:14].rjust(15) for c in cols))
for k in sorted(tab, key=lambda k: -dur.get(k, 0)):
    print(k.ljust(28), f"{dur.get(k, 0):9.1f}", " ".join(f"{tab[k].get(c, 0):15.4g}" for c in cols))
