#!/usr/bin/env python3
"""Instruction histogram of k_ransac<64,16,6> (the benchmarked instantiation) from the compiler's own
assembly: whole kernel, one plane fit (the region up to the first sched_barrier that follows the sampled
points' LDS gathers) and the screened scoring loop.  usage: tools/ransac_isa.py > profiles/rNN_ransac_isa.txt"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "ransac.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
                    f"-I{ROOT}/include", "-S", "--cuda-device-only", f"{ROOT}/octreelib_amd/csrc/ransac.hip", "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN.*k_ransacILi64ELi16ELi6ELi0ELb1ELb1E.*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]


def ops(ls):
    c = collections.Counter()
    for l in ls:
        m = re.match(r"^\s+([a-z_0-9]+)\s", l)
        if m and not l.lstrip().startswith(";"):
            c[m.group(1)] += 1
    return c


def show(title, c):
    tot = sum(c.values())
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(f"== {title}: {tot} instructions, {valu} VALU, {sum(v for k, v in c.items() if k.startswith('ds_'))} LDS, "
          f"{sum(v for k, v in c.items() if k.startswith('s_'))} SALU")
    for k, v in c.most_common(28):
        print(f"   {k:26s}{v}")


show("k_ransac<64,16,6> whole kernel (static)", ops(body))
# one plane fit: from the first run of f64 adds after LDS gathers to the first sched_barrier
sb = [i for i, l in enumerate(body) if "sched_barrier" in l]
# walk back from the barrier over whole basic blocks until the region holds the fit's multiplies
lo = sb[0]
while lo > 0 and sum("v_mul_f64" in l for l in body[lo:sb[0]]) < 70:
    lo -= 1
    while lo > 0 and not body[lo].startswith(".LBB"):
        lo -= 1
fit = body[lo:sb[0]]
show("first plane fit incl. both division / sqrt variants (static; the slow variants are branched over)", ops(fit))
slow = sum(1 for l in fit if re.search(r"v_div_(scale|fmas|fixup)|v_ldexp|v_cmp_class", l))
print(f"   (of these, {slow} belong to the true-division / scaled-sqrt fallbacks that the guards skip)")
# the screened scoring loop: the innermost loop bodies made of v_fma_f32 + v_alignbit
loops = [i for i, l in enumerate(body) if "v_alignbit_b32" in l]
if loops:
    lo, hi = loops[0], loops[0]
    while lo > 0 and not body[lo].startswith(".LBB"):
        lo -= 1
    while hi < len(body) and "s_cbranch" not in body[hi]:
        hi += 1
    show("screened scoring loop, one unrolled body (4 points x 1 hypothesis group)", ops(body[lo:hi + 1]))
