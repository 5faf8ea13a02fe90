#!/usr/bin/env python3
"""Opcode ledger of k_ransac<64,16,6,true,true> (the benchmarked instantiation): every instruction of the compiler's
assembly is attributed to a PHASE of the kernel through its source line (-gline-tables-only: `.loc` directives name
the innermost inlined function's line), classified by opcode, and multiplied by how often its phase runs per launch
(the execution counts of the counting build, profiles/rNN_ransac_counts.json; without that file: static counts only).

    tools/ransac_isa.py [profiles/r06_ransac_counts.json] > profiles/r06_ransac_isa.txt

Phases (source line ranges of csrc/ransac.hip, found by the function / lambda they belong to):
    block     per block: descriptor and point prefetch, staging of the next block, uniforms of the bounds, reduction over
              the wave, winner, outputs, final mask
    fit       one exact plane fit of 64 hypotheses: positions, LDS gathers, plane_from_samples (the reference's f64
              sequence), the screen's constants                         runs: exact groups (group 0 + survivor batches)
    fit_cold  ... its branched-over fall-backs (true division, scaled square root, risky draws)   runs: ~never
    score     the exact count of 64 hypotheses: f32 screen + recount     runs: exact groups; its loop body per 4 points
    pre_fit   approximate f32 plane + bound of 3 x 64 hypotheses          runs: prescreen trios
    pre_screen  their widened-threshold count                             runs: prescreen trios; loop body per 4 points
    queue     survivors -> LDS queue, the batches' bookkeeping            runs: prescreen trios / batches
"""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "octreelib_amd", "csrc", "ransac.hip")
KERNEL = r"^_ZN.*k_ransacILi64ELi16ELi6ELb1ELb1E.*:"

src_lines = open(SRC).read().split("\n")


def line_of(pattern, start=0):
    for i in range(start, len(src_lines)):
        if re.search(pattern, src_lines[i]):
            return i + 1
    raise SystemExit(f"ransac.hip: no line matches {pattern!r}")


# source line ranges -> phase
L = {
    "helpers_begin": 1,
    "plane_fit_begin": line_of(r"^__device__ __forceinline__ double div_by_small_int"),
    "plane_fit_end": line_of(r"^__device__ __forceinline__ double plane_distance"),
    "screen_group": line_of(r"^__device__ __forceinline__ void screen_group"),
    "prescreen_const": line_of(r"^struct PreConst"),
    "screen_ub": line_of(r"^__device__ __forceinline__ void screen_ub"),
    "kernel": line_of(r"^__global__ __launch_bounds__\(THREADS, RS_MINWAVES\) void k_ransac"),
}
L["shared_begin"] = line_of(r"^typedef float f4 ")
L["stage_local"] = line_of(r"^__device__ __forceinline__ void stage_local")
L["load_pos"] = line_of(r"auto load_pos = \[&\]", L["kernel"])
L["fit"] = line_of(r"auto fit = \[&\]", L["kernel"])
L["score"] = line_of(r"auto score = \[&\]", L["kernel"])
L["take"] = line_of(r"auto take = \[&\]", L["kernel"])
L["group0"] = line_of(r"---- group 0", L["kernel"])
L["prescreen"] = line_of(r"---- prescreen of the wave's later hypotheses", L["kernel"])
L["pre_bound_end"] = line_of(r"screen_ub<\w+>\(loc, n", L["kernel"])
L["survivors"] = line_of(r"---- the survivors \(in index order\)", L["kernel"])
L["tail"] = line_of(r"const uint32_t wbest = wave_max_u32\(best\);", L["kernel"])
L["kernel_end"] = line_of(r"^// A block with more than THREADS-1 points", L["kernel"])


def phase_of(line):
    # helpers shared by several phases (fma32, min3abs, the DPP reductions): the phase of the code around them
    if L["shared_begin"] <= line < L["stage_local"] - 4:
        return None
    if L["plane_fit_begin"] <= line < L["plane_fit_end"]:
        return "fit"
    if L["screen_group"] <= line < L["prescreen_const"]:
        return "score"
    if L["prescreen_const"] <= line < L["screen_ub"]:
        return "block"          # prescreen_constants: once per block
    if L["screen_ub"] <= line < L["kernel"]:
        return "pre_screen"
    if L["load_pos"] <= line < L["score"]:
        return "fit"
    if L["score"] <= line < L["take"]:
        return "score"
    if L["take"] <= line < L["group0"]:
        return "fit"            # (take: per exact group)
    if L["group0"] <= line < L["prescreen"]:
        return "fit"
    if L["prescreen"] <= line < L["pre_bound_end"]:
        return "pre_fit"
    if L["pre_bound_end"] <= line < L["survivors"]:
        return "queue"
    if L["survivors"] <= line < L["tail"]:
        return "queue"
    return "block"


def opclass(op):
    if re.match(r"v_(add|mul|fma|fmac|rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|ldexp|max|min|trunc|floor|frexp).*_f64", op):
        return "f64 arithmetic"
    if re.match(r"v_(fma|fmac|mul|add|sub|subrev|pk_add|pk_mul|pk_fma|rsq|rcp|max|min|max3|min3|med3)_f32", op) or op.startswith("v_pk_"):
        return "f32 arithmetic"
    if op.startswith("v_cvt_"):
        return "v_cvt"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "v_cmp"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "v_mov"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane") or "dpp" in op or op.startswith("v_permlane"):
        return "cross-lane"
    if re.match(r"v_(alignbit|bcnt|bfi|bfe|and|or|xor|not|lshl|lshr|ashr|perm)", op):
        return "bit ops (inlier bits, sign, bytes)"
    if re.match(r"v_(add|sub|subrev|mad|mul|lshl_add|add_lshl|lshl_or|and_or|min|max|mbcnt|addc|subb)", op):
        return "integer / address"
    if op.startswith("v_"):
        return "other VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "VMEM"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait / nop"
    if op.startswith("s_"):
        return "SALU"
    return "other"


with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "ransac.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
                    "-gline-tables-only", f"-I{ROOT}/include", "-S", "--cuda-device-only", SRC, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    asm = open(out).read().split("\n")
SRC_FILES = {int(m.group(1)) for m in (re.match(r'^\s+\.file\s+(\d+)\s+.*"(?:[^"]*/)?ransac\.hip"', l) for l in asm) if m}
start = next(i for i, l in enumerate(asm) if re.match(KERNEL, l))
end = next(i for i in range(start, len(asm)) if asm[i].startswith(".Lfunc_end"))
body = asm[start:end]

# instructions with (phase, class, cold?).  Cold = inside a region that a branch skips and that holds the true
# division / scaled square root / exact sample index sequences (never taken on sane data).
COLD = re.compile(r"v_div_scale|v_div_fmas|v_div_fixup|v_ldexp_f64|v_cmp_class|v_frexp|v_cvt_i32_f64|v_sqrt_f64")
blocks, cur = [], []
for l in body:
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append(cur)
        cur = []
    cur.append(l)
blocks.append(cur)
counts = collections.defaultdict(collections.Counter)   # phase -> class -> static count
loop_body = collections.Counter()                       # phase -> VALU instructions inside the 4-point loop bodies
cur_line, cur_file = 0, -1
for b in blocks:
    cold = any(COLD.search(l) for l in b)
    in_loop4 = sum("ds_read_b128" in l for l in b) >= 4 and any("v_alignbit_b32" in l for l in b)
    for l in b:
        m = re.match(r"^\s+\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            # (a line of a runtime header - fma(), __ballot() ... inlined - keeps the phase of the last line of ransac.hip)
            if int(m.group(1)) in SRC_FILES and phase_of(int(m.group(2))) is not None:
                cur_file, cur_line = int(m.group(1)), int(m.group(2))
            continue
        m = re.match(r"^\s+([a-z_0-9]+)(\s|$)", l)
        if not m or l.lstrip().startswith(";") or l.lstrip().startswith("."):
            continue
        op = m.group(1)
        ph = phase_of(cur_line) if cur_file in SRC_FILES else "block"
        if cold and ph == "fit":
            ph = "fit_cold"
        counts[ph][opclass(op)] += 1
        if in_loop4 and op.startswith("v_"):
            loop_body[ph] += 1

phases = ["block", "fit", "fit_cold", "score", "pre_fit", "pre_screen", "queue"]
classes = ["f64 arithmetic", "f32 arithmetic", "v_cvt", "v_cmp", "v_cndmask", "v_mov", "cross-lane",
           "bit ops (inlier bits, sign, bytes)", "integer / address", "other VALU", "LDS", "VMEM", "SALU", "wait / nop"]
VALU = classes[:10]
print("k_ransac<64,16,6,true,true>: static instruction counts by phase and opcode class")
print("(" + ", ".join(f"{k} = line {v}" for k, v in sorted(L.items(), key=lambda kv: kv[1]) if k not in ("helpers_begin",)) + ")")
print()
print("class".ljust(38) + "".join(p.rjust(12) for p in phases))
for c in classes:
    print(c.ljust(38) + "".join(str(counts[p][c]).rjust(12) for p in phases))
print("VALU total".ljust(38) + "".join(str(sum(counts[p][c] for c in VALU)).rjust(12) for p in phases))
print("  of which in a 4-point loop body".ljust(38) + "".join(str(loop_body[p]).rjust(12) for p in phases))

cnt = None
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    cnt = json.load(open(sys.argv[1]))
if cnt:
    blocks_n = cnt["blocks_per_launch"]
    nbar = cnt["mean_block_size"]
    exact_groups = cnt["plane_fits_executed_exactly"] / 64.0
    trios = cnt["hypotheses_prescreened"] / 64.0 / 3.0
    batches = cnt["survivor_batches"]
    # runs per launch of each phase's straight-line part; loop bodies run once per 4 points of a block
    runs = {"block": blocks_n, "fit": exact_groups, "fit_cold": 0.0, "score": exact_groups, "pre_fit": trios / 1.0,
            "pre_screen": trios, "queue": trios + batches}
    # pre_fit's static count holds the THREE unrolled fits of a trio: one run per trio
    loops = {"score": exact_groups * nbar / 4.0, "pre_screen": trios * nbar / 4.0}
    print()
    print(f"dynamic estimate per launch (wave-level VALU instructions, millions): {blocks_n:.0f} blocks of {nbar:.1f} points, "
          f"{exact_groups / blocks_n:.2f} exact groups and {trios / blocks_n:.2f} prescreen trios per block")
    print("class".ljust(38) + "".join(p.rjust(12) for p in phases) + "total".rjust(12))
    tot_all = 0.0
    for c in VALU:
        row, tot = [], 0.0
        for p in phases:
            st = counts[p][c]
            v = st * runs[p]
            if p in loops and loop_body[p]:
                # the class's share of the loop body runs with the loop, the rest with the phase
                frac = loop_body[p] / max(1, sum(counts[p][k] for k in VALU))
                v = st * (1 - frac) * runs[p] + st * frac * loops[p]
            row.append(v)
            tot += v
        tot_all += tot
        print(c.ljust(38) + "".join(f"{v / 1e6:12.1f}" for v in row) + f"{tot / 1e6:12.1f}")
    print("VALU total".ljust(38) + " " * (12 * len(phases)) + f"{tot_all / 1e6:12.1f}")
    print("(the loop-body share of a class is taken as the phase's average: an estimate, high on the classes that sit "
          "outside the loops)")
    # the counters' own totals beside the estimate (tools/profile_round.sh: the SQ pass and the VALU-mix pass)
    tag = os.path.basename(sys.argv[1]).split("_")[0]
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters.json")))["kernels"]
        mix = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_valu_mix.json")))["kernels"]
        key = next(k for k in sq if k.startswith("k_ransac<64"))
        m = mix[key]
        f64 = m["add_f64"] + m["mul_f64"] + m["fma_f64"] + m["trans_f64"]
        f32 = m["add_f32"] + m["mul_f32"] + m["fma_f32"]
        allv = sq[key]["wave_valu_instructions"]
        print()
        print(f"MEASURED per launch (profiles/{tag}_sq_counters.json, {tag}_valu_mix.json): all VALU {allv / 1e6:.1f} M wave "
              f"instructions; f64 {f64 / 1e6:.1f} M, f32 {f32 / 1e6:.1f} M, int32 {m['int32'] / 1e6:.1f} M, everything else "
              f"{(allv - f64 - f32 - m['int32']) / 1e6:.1f} M.  Estimate / measured: all {tot_all / allv:.2f}")
    except (OSError, KeyError, StopIteration):
        pass
