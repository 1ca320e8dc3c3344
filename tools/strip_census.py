#!/usr/bin/env python3
"""
ISA census of k_visualizer_strip<72, 12, 2, 9, 6, 4, false>: instructions per supersample BY CLASS, priced with
the issue cycles tools/ubench_valu.hip measured, so that bench.py's `issue_model` prices the 16 % of the VALU instructions no hardware
class counter covers ("other": moves, selects, compares, min/max, fract/floor, readfirstlane) with what they are instead of a 2-or-4 band.

Method. `tools/isa_census.py LISTING KERNEL --dump blocks.json` (LISTING built with -gline-tables-only) gives every basic block's
instructions by class and the source lines it was generated from. The kernel's control flow is SCALAR (the lanes of a wave share their
rows): which blocks run is decided by per-frame tables, and the unrolled copies of a phase are identical, so a block's executions per
wave follow from a handful of event rates:
    folds      diagonal cell folds per wave — MEASURED by the profiling build (-DSF_SECTION_TIMERS counts them: 53.0 of 180 advances at C3);
               the first advance of every (step, side) always folds (20), the other 160 fold with p = (folds - 20)/160; the row-line's
               P/Q fold follows the same statistic (a new row of cells under the next sample): 1 + 8 p per strip
    columns    iterations of the column-line loop: 8 slots + the distinct cell rows under the strip - 1 = 8 + 8 p
    exact      waves that re-run the polar chain exactly (speculation margin): 2 % (1-3 % of the row-waves)
    far/bars/strips  divergent colour branches of visualizer.frag:49-73 (a wave runs a branch when ANY lane takes it): estimates, small blocks
The model's per-class totals are then CHECKED against the hardware's class counters of the same launch (SQ_INSTS_VALU_*): the split of
"other" is credible to the extent the counted classes agree.

usage: tools/strip_census.py blocks.json [bench.json with roofline.issue_model] [--folds 53.0]
"""
from __future__ import annotations

import argparse
import collections
import json
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from isa_census import CLASSES, VALU  # noqa: E402

CYCLES = {name: cycles for name, _, cycles in CLASSES}
# the files k_visualizer_strip is compiled from (bench.py ties the census to them, not to the whole library)
STRIP_SOURCES = ("visualizer_fast.hpp", "visualizer_kernels.hpp", "render_kernels.hpp", "fragments.hpp", "glsl.hpp", "sfmath.hpp", "launch_visualizer_strip.hip", "Makefile")


def vf_lines(block) -> collections.Counter:
    out: collections.Counter = collections.Counter()
    for line, count in block["lines"].items():
        name, _, number = line.partition(":")
        if name == "visualizer_fast.hpp" and number.isdigit():
            out[int(number)] += count
    return out


def other_lines(block) -> collections.Counter:
    out: collections.Counter = collections.Counter()
    for line, count in block["lines"].items():
        if not line.startswith("visualizer_fast.hpp"):
            out[line] += count
    return out


def source_ranges(path: str) -> dict:
    """Line ranges of the kernel's phases in visualizer_fast.hpp, from the markers in its text (the listing's .loc lines refer to them)"""
    text = open(path).read().split("\n")
    def find(needle: str, after: int = 0) -> int:
        for number, line in enumerate(text, 1):
            if number > after and needle in line:
                return number
        raise SystemExit(f"marker '{needle}' not found in {path}")
    strip = find("struct VisualizerStrip {")
    marks = {
        "post_fn": (find("__device__ __forceinline__ uint32_t visualizer_fast_post("), find("// ---- the pixel tier (round 6)")),
        "gains_fn": (find("__device__ __forceinline__ PixelGains visualizer_pixel_gains("), find("// One thread per (frame, wave tile)")),
        "tier_fn": (find("template <int SOURCE> __device__ __forceinline__ static float from_pixel_lane(", strip), find("__device__ static void run(const RenderArgs& a, const VisTables& t) {", strip)),
        "stage_fn": (find("__device__ __forceinline__ void visualizer_fast_stage("), find("// ---- the fused kernel ----")),
        "blur_direct": (find("__device__ static void blur_direct("), find("__device__ static void run(const RenderArgs& a, const VisTables& t) {")),
        "load_cell": (find("__device__ __forceinline__ static void load_cell(", strip), find("template <class T> __device__ __forceinline__ static float F(", strip)),
        "rowline": (find("// ---- the row-lines:", strip), find("// row-lines", strip)),
        "colline": (find("// ---- the column-lines:", strip), find("// column-lines", strip)),
        "diag": (find("// ---- the four diagonal directions:", strip), find("// diagonals", strip)),
        "post": (find("// ---- visualizer.frag:36-73 per sample", strip), find("// post-processing", strip)),
        "resolve": (find("// every wave is done with the cells", strip), find("// barrier + texel exchange + resolve + barrier", strip)),
        "diag_setup_end": (find("for (int w = 0; w < 10; w++) {", find("// ---- the four diagonal directions:", strip)),)*2,
        "advance_use": (find("acc[r][2] = acc[r][2] + U[side][2];", strip),)*2,
        "pow_line": (find("col = col*ColourMath<true>::pow((len - rr)*0.5f, 0.05f);"),)*2,
        "mix_line": (find("if (len < rr) col = mix(col, vec3{1.0f, 1.0f, 1.0f}, smoothstep01(0.5f + bar));"),)*2,
        "strips_line": (find("if (strip_top) { col = col*0.8f; opacity = opacity*0.8f; }"),)*2,
        "smoothstep01": (0, 0),      # (round 6: moved to render_kernels.hpp — the mix branch is recognised by its own line alone)
        "store": (find("// barrier + texel exchange + resolve + barrier", strip), find("template <int TILE_PITCH, int TILE_ROWS, int S, int WALK, int MIN_WAVES", strip)),
        "sweep": (find("constexpr int GROUPS = BLOCK_PX*3/16;", strip), find("// the sweep of stores", strip)),
    }
    return marks


def within(lines, span) -> bool:
    return any(span[0] <= line <= span[1] for line in lines)


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--source", default=__file__.rsplit("/", 2)[0] + "/shaderflow_amd/csrc/visualizer_fast.hpp")
    parser.add_argument("blocks")
    parser.add_argument("bench", nargs="?")
    parser.add_argument("--folds", type=float, default=53.0, help="diagonal folds per wave measured by the -DSF_SECTION_TIMERS build")
    parser.add_argument("--exact", type=float, default=0.02)
    parser.add_argument("--per-sample", type=float, default=0.145, help="share of the waves that evaluate visualizer.frag:36-73 per sample (round 6: the others take the pixel tier); "
                        "measured: sfx_ctx_tile_misses over a bench run / (frames x 57 600 waves)")
    parser.add_argument("--strip-waves", type=float, default=0.215, help="share of the waves that can see a waveform strip (the tier evaluates its two compares per sample only there)")
    parser.add_argument("--weights-out", help="write [label, weight, why] per block as JSON (to look at a phase's blocks by hand)")
    parser.add_argument("--json", help="write the prices bench.py's issue_model reads (profiles/r05_strip_isa_census.json)")
    args = parser.parse_args()
    blocks = json.load(open(args.blocks))
    M = source_ranges(args.source)
    p = (args.folds - 20.0)/160.0
    rates = {"fold_first": 1.0, "fold_next": p, "row_folds": 1.0 + 8.0*p, "column_iterations": 8.0 + 8.0*p, "exact": args.exact,
             "pow_branch": 0.95, "mix_branch": 0.30, "strips": 0.06, "staging_rounds": 864.0/512.0, "resolve_rounds": 2.25,
             "per_sample_waves": args.per_sample, "tier_waves": 1.0 - args.per_sample, "tier_strip_waves": args.strip_waves}

    # ---- a weight per block: walk the listing in order, phase by phase ------------------------------------------------------------
    weights = [0.0]*len(blocks)
    why = [""]*len(blocks)
    phase = "prologue"
    diag_fold_seen = 0
    row_fold_seen = 0
    for i, block in enumerate(blocks):
        n = sum(block["counts"].values())
        lines, others = vf_lines(block), other_lines(block)
        top = lines.most_common(1)[0][0] if lines else 0
        if phase == "prologue" and within(lines, M["rowline"]):
            phase = "rowline"
        if phase == "rowline" and within(lines, M["colline"]):
            phase = "colline"
        if phase == "colline" and within(lines, (M["diag"][0], M["diag_setup_end"][0] + 6)):
            phase = "diag"
        if phase == "diag" and within(lines, M["post"]):
            phase = "post"
        if phase == "post" and within(lines, M["resolve"]) and not within(lines, M["post_fn"]):
            phase = "resolve"
        if phase == "resolve" and within(lines, (M["store"][0] + 1, M["store"][1])) and not any(line <= M["store"][0] for line in lines):
            phase = "store"

        tier_lines = (within(lines, M["gains_fn"]) or within(lines, M["tier_fn"])) and not within(lines, M["post_fn"]) and not within(lines, (M["rowline"][0], M["store"][1]))
        if tier_lines and phase != "prologue":
            # the pixel tier (wherever the compiler laid its blocks out: they sit behind the store phase in the listing): two instances — the
            # wave's first row modulo 2 —, each run by half of the tier's waves
            text = " ".join(others)
            share = 0.5*rates["tier_waves"]
            if "glsl.hpp:81" in text or "glsl.hpp:82" in text:
                weights[i], why[i] = 0.0, "wrap_texel's modulo (bin outside the texture: never)"
            elif n <= 8 and block["counts"].get("mul_f32", 0) >= 1 and block["counts"].get("cmp", 0) + block["counts"].get("cndmask", 0) >= 1 and not block["counts"].get("cvt"):
                weights[i], why[i] = share*rates["tier_strip_waves"], "pixel tier: waveform strips"
            else:
                weights[i], why[i] = share, "pixel tier"
            continue
        if phase == "prologue":
            if within(lines, M["blur_direct"]) or (n in (83, 84) and "vmem" in block["counts"]):
                weights[i], why[i] = 0.0, "blur_direct (a window off its tile: never at C3)"
            elif any("glsl.hpp:8" in line for line in others) and not lines:
                weights[i], why[i] = rates["staging_rounds"]*0.02, "wrap_texel's modulo (texels outside the texture: the frame's rim)"
            elif within(lines, M["stage_fn"]):
                weights[i], why[i] = rates["staging_rounds"], "staging loop"
            else:
                weights[i], why[i] = 1.0, "prologue"
            # the blur_direct chain's helper blocks (glsl.hpp wrap, its loop control) sit between its bodies: they run only with it
            if 2 < i and why[i] == "prologue" and not lines and any(k.startswith("glsl.hpp") for k in others):
                weights[i], why[i] = 0.0, "blur_direct helper"
        elif phase == "rowline":
            if n > 100:
                row_fold_seen += 1
                weights[i] = 1.0 if row_fold_seen == 1 else p
                why[i] = "row-line P/Q fold"
            else:
                weights[i], why[i] = 1.0, "row-line per sample"
        elif phase == "colline":
            if block["targets"] and block["targets"][0] == block["label"]:
                weights[i], why[i] = rates["column_iterations"], "column-line loop body"
            else:
                weights[i], why[i] = 1.0, "column-line set-up"
        elif phase == "diag":
            fold = 28 <= n <= 31 and block["counts"].get("lds", 0) == 6
            skip_path = (not block["counts"].get("lds")) and all(cls in ("salu", "branch", "mov", "waitcnt") for cls in block["counts"]) and n <= 14 and not within(lines, (M["diag"][0], M["diag_setup_end"][0] + 12)) and i > 0 and "diagonal set-up" not in why[i - 1]
            if block["counts"].get("lds", 0) >= 12 and n > 40:
                # VIS_STRIP_FIRST_ROW_FOLDS: the first row's two folds (twelve reads) sit in the step's own block; every other fold is conditional
                diag_fold_seen = 2
                weights[i], why[i] = 10.0, "diagonal per step / per advance"
            elif fold:
                diag_fold_seen += 1
                weights[i] = 10.0*(1.0 if diag_fold_seen <= 2 else p)
                why[i] = "diagonal fold"
            elif within(lines, (M["diag"][0], M["diag_setup_end"][0])) and n > 60:
                weights[i], why[i] = 1.0, "diagonal set-up"
            elif skip_path and re.fullmatch(r"\.LBB\d+_\d+", block["label"]) and block["counts"].get("salu", 0) >= 2 and not block["counts"].get("smem"):
                weights[i], why[i] = 0.0, "rows beyond the frame (never inside it)"
            elif block["counts"].get("mov", 0) >= 10 and n <= 18:
                weights[i], why[i] = 0.0, "rows beyond the frame (zeroed sums)"
            else:
                weights[i], why[i] = 10.0, "diagonal per step / per advance"
        elif phase == "post":
            text = " ".join(others)
            per_sample = rates["per_sample_waves"] if within(lines, M["post_fn"]) or any(k.startswith(("fragments.hpp", "sfmath.hpp", "__clang_hip_math.h")) for k in others) else 1.0
            if "glsl.hpp:81" in text or "glsl.hpp:82" in text:
                weights[i], why[i] = 9.0*0.0, "wrap_texel's modulo (bin outside the texture: never)"
            elif "sfmath.hpp:110" in text or "sfmath.hpp:99" in text or "glsl.hpp:403" in text or "__clang_hip_math.h:722" in text:
                weights[i], why[i] = rates["exact"]*rates["per_sample_waves"], "exact polar chain (speculation re-run)"
            elif M["pow_line"][0] in lines or "fragments.hpp:0" in text:
                weights[i], why[i] = rates["pow_branch"]*rates["per_sample_waves"], "pow((len - r)/2, 0.05): lanes beyond their bar"
            elif M["mix_line"][0] in lines and block["counts"].get("fma_f32", 0) >= 3 and n <= 24:
                weights[i], why[i] = rates["mix_branch"]*rates["per_sample_waves"], "mix towards white: lanes inside a bar"
            elif M["strips_line"][0] in lines or (top == 0 and n <= 3 and i > 0 and "strips" in why[i - 1]):
                weights[i], why[i] = rates["strips"]*rates["per_sample_waves"], "waveform strips / out-of-aspect bars"
            else:
                weights[i], why[i] = per_sample, ("post (per-sample waves)" if per_sample < 1.0 else "post")
        elif phase == "resolve":
            loop = within(lines, (M["resolve"][0] + 6, M["resolve"][1] - 2)) or any("render_kernels.hpp:30" in k or "render_kernels.hpp:28" in k or "render_kernels.hpp:0" in k for k in others)
            weights[i], why[i] = (rates["resolve_rounds"] if loop else 1.0), "texel exchange + resolve"
        else:
            sweep = within(lines, M["sweep"])
            weights[i], why[i] = (1.0 if sweep or within(lines, (M["store"][0], M["sweep"][0] + 2)) else 0.0), ("sweep store" if sweep else "store (row-by-row fallback: not at C3)")

    # two column loops in the listing = the aligned-table loop and its fallback (a strip spanning more cells than the table holds: never
    # at C3): the fallback does not run
    loops = [i for i, reason in enumerate(why) if reason == "column-line loop body"]
    if len(loops) > 1:
        keep = min(loops, key=lambda i: blocks[i]["counts"].get("salu", 0))
        for i in loops:
            if i != keep:
                weights[i], why[i] = 0.0, "column-line fallback loop (never at C3)"

    if args.weights_out:
        json.dump([[b["label"], w, r] for b, w, r in zip(blocks, weights, why)], open(args.weights_out, "w"))

    # ---- totals ---------------------------------------------------------------------------------------------------------------------
    by_phase: dict[str, collections.Counter] = collections.defaultdict(collections.Counter)
    total: collections.Counter = collections.Counter()
    sgpr = 0.0
    sgpr_by_phase: collections.Counter = collections.Counter()
    for block, weight, reason in zip(blocks, weights, why):
        group = reason.split(" (")[0].split(":")[0]
        for cls, count in block["counts"].items():
            total[cls] += count*weight
            by_phase[group][cls] += count*weight
        sgpr += block["sgpr_forms"]*weight
        sgpr_by_phase[group] += block["sgpr_forms"]*weight
    samples = 9.0
    valu_total = sum(total[c] for c in VALU)
    print(f"event rates: {json.dumps({k: round(v, 4) for k, v in rates.items()})}")
    print(f"\nwave-instructions per wave (576 supersamples) -> per supersample (x 64 lanes / 576 = / 9):")
    print(f"  {'class':16s} {'per wave':>10s} {'per sample':>11s} {'cycles':>7s} {'issue cycles / sample':>22s}")
    cycles_total = 0.0
    for name, _, cycles in CLASSES:
        if total[name] == 0:
            continue
        print(f"  {name:16s} {total[name]:10.1f} {total[name]/samples:11.2f} {cycles:7d} {total[name]/samples*cycles:22.2f}")
        if name in VALU:
            cycles_total += total[name]/samples*cycles
    print(f"  {'SGPR-source forms':16s} {sgpr:10.1f} {sgpr/samples:11.2f} {'+2':>7s} {sgpr/samples*2:22.2f}")
    cycles_total += sgpr/samples*2
    print(f"  VALU: {valu_total:.0f} per wave = {valu_total/samples:.1f} per supersample; {cycles_total:.1f} issue cycles per supersample = {cycles_total/(valu_total/samples):.3f} cycles per VALU instruction")
    print(f"  scalar: {total['salu']/samples:.1f} SALU + {total['smem']/samples:.1f} SMEM + {total['branch']/samples:.1f} branches + {total['waitcnt']/samples:.1f} waits per supersample; LDS {total['lds']/samples:.1f}, VMEM {total['vmem']/samples:.2f}")
    print("\nby phase (VALU instructions per supersample | share | of which moves, compares, selects, integer | LDS, SALU):")
    for group, counts in sorted(by_phase.items(), key=lambda kv: -sum(kv[1][c] for c in VALU)):
        v = sum(counts[c] for c in VALU)
        if v > 0.5:
            print(f"  {group:40s} {v/samples:8.1f}  {100*v/valu_total:5.1f} %   mov {counts['mov']/samples:5.1f} cmp {counts['cmp']/samples:5.1f} cndmask {counts['cndmask']/samples:5.1f} "
                  f"int {(counts['int_shift_mul'] + counts['int_add_logic'])/samples:5.1f} sgpr-src {sgpr_by_phase[group]/samples:5.1f} pk {(counts['pk_fma_f32'] + counts['pk_add_f32'] + counts['pk_mul_f32'])/samples:5.1f}   lds {counts['lds']/samples:5.1f} salu {counts['salu']/samples:5.1f}")

    # (the packed forms are counted ONCE by their plain form's class counter: r05_bench_c3 lost 20 adds and 20 fmas per supersample when
    # forty of each became twenty packed ones)
    hw_map = {"fma_f32": ["fma_f32", "pk_fma_f32"], "add_f32": ["add_f32", "pk_add_f32"], "mul_f32": ["mul_f32", "pk_mul_f32"], "trans_f32": ["trans"], "cvt": ["cvt"],
              "int32": ["int_shift_mul", "int_add_logic"]}
    other_classes = [c for c in VALU if not any(c in v for v in hw_map.values())]
    if args.bench:
        record = json.loads([line for line in open(args.bench) if line.startswith("{")][-1])
        model = record["roofline"]["issue_model"]
        frames = record["roofline"]["frames_per_launch"]
        waves = frames*30*240*8.0
        print(f"\ncheck against the hardware's class counters of the same launch ({args.bench}; wave-instructions per wave):")
        print(f"  {'class':12s} {'model':>9s} {'measured':>9s} {'model/measured':>15s}")
        for hw, mine in hw_map.items():
            modelled = sum(total[c] for c in mine)
            measured = model["instructions"][hw]/waves
            print(f"  {hw:12s} {modelled:9.1f} {measured:9.1f} {modelled/measured:15.3f}")
        modelled_other = sum(total[c] for c in other_classes)
        print(f"  {'other':12s} {modelled_other:9.1f} {model['instructions']['other']/waves:9.1f} {modelled_other/(model['instructions']['other']/waves):15.3f}")
        print(f"  {'all':12s} {valu_total:9.1f} {model['instructions']['all']/waves:9.1f} {valu_total/(model['instructions']['all']/waves):15.3f}")
    other_cycles = sum(total[c]*CYCLES[c] for c in other_classes)
    other_count = sum(total[c] for c in other_classes)
    print(f"\n'other' (no hardware class counter): {other_count/samples:.1f} per supersample = " + ", ".join(f"{c} {total[c]/samples:.1f}" for c in other_classes if total[c] > 0.05*samples))
    print(f"  priced by class: {other_cycles/other_count:.3f} issue cycles per instruction (bench.py issue_model: OTHER_CYCLES)")
    if args.json:
        import hashlib
        from pathlib import Path
        csrc = Path(args.source).parent
        digest = hashlib.sha256()
        for name in STRIP_SOURCES:
            digest.update(name.encode()); digest.update((csrc/name).read_bytes())
        json.dump({"kernel": "k_visualizer_strip<72, 12, 2, 9, 6, 4, false>", "strip_sources_fingerprint": digest.hexdigest()[:16],
                   "other_cycles_per_instruction": round(other_cycles/other_count, 4),
                   "sgpr_source_full_rate_forms_per_valu_instruction": round(sgpr/valu_total, 5),
                   "packed_f32_forms_per_valu_instruction": round((total["pk_fma_f32"] + total["pk_add_f32"] + total["pk_mul_f32"])/valu_total, 5),
                   "valu_instructions_per_supersample_modelled": round(valu_total/samples, 2),
                   "average_issue_cycles_per_valu_instruction": round(cycles_total/(valu_total/samples), 4),
                   "other_by_class_per_supersample": {c: round(total[c]/samples, 2) for c in other_classes if total[c] > 0},
                   "event_rates": rates, "how": "tools/strip_census.py on tools/isa_census.py --dump of the -gline-tables-only listing; prices: profiles/r02_ubench_valu.txt, r05_ubench_valu_sgpr.txt, r05_ubench_pk_f32.txt"},
                  open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
