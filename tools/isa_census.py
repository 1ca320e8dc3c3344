#!/usr/bin/env python3
"""
ISA census of one kernel of `make asm`'s output (shaderflow_amd/csrc/launch_visualizer_strip.gfx950.s): the kernel's basic blocks, the instructions of
every block by CLASS (the classes tools/ubench_valu.hip prices: full-rate f32 add / mul / fma and moves, half-rate conversions /
min-max / fract / shifts / SGPR-operand forms, quarter-rate transcendentals; scalar ALU, scalar loads, branches, waits, LDS, global),
the source lines a block was generated from (when the listing carries `.loc`: build it with -gline-tables-only), and — given how often
every block runs — instructions per supersample by class, priced in SIMD cycles.

How often a block runs comes from one of
  --weights FILE.json     {"<label>": executions per wave, …} (e.g. from PC samples: tools/pc_samples_to_weights.py), or
  --model strip           the analytic trip counts of k_visualizer_strip at a given geometry (the scalar branches of that kernel are
                          decided by the per-frame tables alone: `same row of cells as the previous sample?`), see strip_model().

usage: tools/isa_census.py LISTING.s KERNEL_SUBSTRING [--weights w.json | --model strip] [--blocks] [--samples N]
"""
from __future__ import annotations

import argparse
import collections
import json
import re
import sys

# cycles a wave64 instruction of the class occupies its SIMD's issue port (profiles/r02_ubench_valu.txt, r05_ubench_valu_sgpr.txt, at
# the sustained clock): 2 for the f32 add / mul / fma, moves and integer adds between VECTOR registers; 4 for conversions, min / max /
# med3, fract / floor, shifts, compares, readfirstlane, SDWA — and for ANY form with a scalar-register source (v_mov_b32 v, s included);
# 8 for the transcendentals
CLASSES = [
    # (class, regex on the mnemonic, cycles)
    ("fma_f32", r"v_(fma|fmac|fmaak|fmamk|mad|mac)_f32", 2),
    ("add_f32", r"v_(add|sub|subrev)_f32", 2),
    ("mul_f32", r"v_mul_f32", 2),
    ("pk_fma_f32", r"v_pk_fma_f32", 4),                               # the packed forms cost two plain ones whatever their sources (profiles/r05_ubench_pk_f32.txt);
    ("pk_add_f32", r"v_pk_add_f32", 4),                               # the hardware counts each ONCE, in its plain form's class counter
    ("pk_mul_f32", r"v_pk_mul_f32", 4),
    ("fma_mix", r"v_fma_mix", 4),
    ("trans", r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_(f32|iflag_f32|legacy_f32)", 8),
    ("cvt", r"v_cvt_", 4),
    ("fract_floor", r"v_(fract|floor|ceil|rndne|trunc)_f32", 4),
    ("minmax_med", r"v_(min|max|med3|min3|max3)_", 4),
    ("cmp", r"v_cmpx?_", 4),                                          # 4.3 measured, into vcc or into a scalar pair (profiles/r05_ubench_valu_sgpr.txt)
    ("cndmask", r"v_cndmask_", 2),
    ("mov", r"v_(mov_b32|mov_b64|accvgpr)", 2),
    ("dpp_swizzle", r"v_(mov_b32_dpp|permlane|swap)", 4),
    ("readlane", r"v_(readfirstlane|readlane|writelane)", 4),
    ("int_shift_mul", r"v_(lshl|lshr|ashr|mul_lo|mul_hi|mad_u|mad_i|mul_u32|mul_i32|bfe|bfi|perm|lshl_add|lshl_or|and_or|or3|add3|add_lshl|alignbit)", 4),
    ("int_add_logic", r"v_(add|sub|subrev|addc|subb)_(u|i|co)|v_(and|or|xor|not)_b", 2),
    ("valu_other", r"v_", 4),
    ("lds", r"ds_", 0),
    ("vmem", r"(global|buffer|flat|scratch)_", 0),
    ("smem", r"s_(load|buffer_load)", 0),
    ("branch", r"s_(cbranch|branch|setpc|swappc|endpgm)", 0),
    ("waitcnt", r"s_waitcnt", 0),
    ("barrier", r"s_barrier", 0),
    ("nop_sleep", r"s_(nop|sleep|setprio|sethalt)", 0),
    ("salu", r"s_", 0),
]
VALU = [name for name, pattern, _ in CLASSES if pattern.startswith("v_") or name in ("dpp_swizzle", "readlane", "int_shift_mul", "int_add_logic", "valu_other")]
COMPILED = [(name, re.compile(pattern), cycles) for name, pattern, cycles in CLASSES]


def classify(mnemonic: str, operands: str) -> str:
    if mnemonic.endswith("_dpp") or " row_" in operands or "quad_perm" in operands:
        return "dpp_swizzle"
    for name, pattern, _ in COMPILED:
        if pattern.match(mnemonic):
            return name
    return "other"


def uses_sgpr_source(mnemonic: str, operands: str) -> bool:
    """VALU forms with a scalar register as a SOURCE issue at half rate (ubench: v_fma_f32 with an SGPR operand 4.4 cycles)"""
    if not mnemonic.startswith("v_") or mnemonic.startswith(("v_readfirstlane", "v_readlane", "v_cmp")):
        return False
    parts = [p.strip() for p in operands.split(",")]
    return any(re.match(r"^-?\|?(s\d+|s\[\d+:\d+\]|vcc|exec)", p) for p in parts[1:])


class Block:
    def __init__(self, label: str):
        self.label = label
        self.counts: collections.Counter = collections.Counter()
        self.sgpr_forms = 0
        self.lines: collections.Counter = collections.Counter()
        self.instructions: list[tuple[str, str]] = []
        self.targets: list[str] = []
        self.offset = 0                                               # byte offset of the block inside the kernel (8 bytes assumed for VOP3/literals: approximate)


def parse(listing: str, kernel: str) -> tuple[str, list[Block]]:
    blocks: list[Block] = []
    inside = False
    name = ""
    files: dict[str, str] = {}
    current_line = ""
    with open(listing) as handle:
        for raw in handle:
            if not inside:
                match = re.match(r"^(\S+):\s*; @", raw)
                if match and kernel in match.group(1) and not match.group(1).startswith("."):
                    inside, name = True, match.group(1)
                    blocks.append(Block("entry"))
                elif raw.startswith("\t.file\t"):
                    f = re.match(r'\t\.file\t(\d+) "([^"]*)"(?: "([^"]*)")?', raw)
                    if f:
                        files[f.group(1)] = (f.group(3) or f.group(2)).split("/")[-1]
                continue
            if raw.startswith(".Lfunc_end"):
                break
            label = re.match(r"^(\.LBB\d+_\d+):", raw)
            if label:
                blocks.append(Block(label.group(1)))
                continue
            fall = re.match(r"^; %bb\.(\d+):", raw)
            if fall and blocks[-1].instructions:
                blocks.append(Block(f"bb.{fall.group(1)}"))
                continue
            if raw.startswith("\t.loc\t"):
                loc = raw.split()
                current_line = f"{files.get(loc[1], loc[1])}:{loc[2]}"
                continue
            if raw.startswith("\t.file\t"):
                f = re.match(r'\t\.file\t(\d+) "([^"]*)"(?: "([^"]*)")?', raw)
                if f:
                    files[f.group(1)] = (f.group(3) or f.group(2)).split("/")[-1]
                continue
            if not raw.startswith("\t") or raw.startswith("\t.") or raw.startswith("\t;"):
                continue
            text = raw.split(";")[0].strip()
            if not text:
                continue
            mnemonic, _, operands = text.partition(" ")
            mnemonic = re.sub(r"_e(32|64)(_dpp)?$", lambda m: m.group(2) or "", mnemonic)
            block = blocks[-1]
            kind = classify(mnemonic, operands)
            block.counts[kind] += 1
            if kind in VALU and uses_sgpr_source(mnemonic, operands) and dict((n, c) for n, _, c in CLASSES).get(kind) == 2:
                block.sgpr_forms += 1                                 # full-rate forms pay for a scalar source; half-rate ones already take 4
            block.instructions.append((mnemonic, operands))
            if current_line:
                block.lines[current_line] += 1
            if kind == "branch":
                target = re.search(r"(\.LBB\d+_\d+)", operands)
                if target:
                    block.targets.append(target.group(1))
    return name, blocks


def priced(counts: collections.Counter, sgpr_forms: float = 0.0) -> float:
    cycles = {name: c for name, _, c in CLASSES}
    total = sum(counts[k]*cycles.get(k, 0) for k in counts)
    return total + 2.0*sgpr_forms                                     # an SGPR source doubles a full-rate form (2 → 4); half-rate forms stay


def report(name: str, blocks: list[Block], weights: dict[str, float] | None, samples_per_wave: float, show_blocks: bool) -> None:
    print(f"kernel {name}")
    print(f"{len(blocks)} basic blocks, {sum(sum(b.counts.values()) for b in blocks)} instructions in the listing "
          f"({sum(sum(b.counts[k] for k in VALU) for b in blocks)} VALU)")
    if show_blocks:
        for b in blocks:
            top = ", ".join(f"{line} x{n}" for line, n in b.lines.most_common(3))
            valu = sum(b.counts[k] for k in VALU)
            w = f"{weights.get(b.label, 0.0):9.3f}" if weights is not None else "        -"
            print(f"  {b.label:12s} runs/wave {w}  {sum(b.counts.values()):4d} instr ({valu:4d} VALU, {b.counts['salu']:3d} SALU, {b.counts['lds']:3d} LDS, "
                  f"{b.counts['smem']:2d} SMEM, {b.counts['waitcnt']:2d} waits) -> {','.join(b.targets) or '-'}   [{top}]")
    if weights is None:
        return
    total: collections.Counter = collections.Counter()
    sgpr = 0.0
    for b in blocks:
        w = weights.get(b.label, 0.0)
        for k, n in b.counts.items():
            total[k] += n*w
        sgpr += b.sgpr_forms*w
    per = 1.0/samples_per_wave
    cycles = {name: c for name, _, c in CLASSES}
    print(f"\nper supersample (one wave = 64 lanes x {samples_per_wave/64:g} samples):")
    print(f"  {'class':16s} {'instr':>9s} {'cycles each':>12s} {'SIMD cycles':>12s}")
    valu_total = valu_cycles = 0.0
    for k, _, c in CLASSES:
        if total[k] == 0:
            continue
        print(f"  {k:16s} {total[k]*per*64:9.2f} {c:12d} {total[k]*per*64*c:12.2f}")
        if k in VALU:
            valu_total += total[k]*per*64
            valu_cycles += total[k]*per*64*c
    print(f"  {'(SGPR-source forms)':16s} {sgpr*per*64:9.2f} {'+2':>12s} {sgpr*per*64*2:12.2f}")
    valu_cycles += sgpr*per*64*2
    print(f"  VALU instructions per supersample (lane-instructions / 64 lanes x 64): {valu_total:.1f}; issue cycles per supersample-wave-slot: {valu_cycles:.1f}"
          f" = {valu_cycles/valu_total:.3f} cycles per VALU instruction on average")
    print(f"  scalar ALU per VALU instruction: {total['salu']/max(1e-9, sum(total[k] for k in VALU)):.3f}; waits {total['waitcnt']*per*64:.2f}, LDS {total['lds']*per*64:.2f}, "
          f"SMEM {total['smem']*per*64:.2f}, branches {total['branch']*per*64:.2f} per supersample")


def main() -> None:
    p = argparse.ArgumentParser()
    p.add_argument("listing")
    p.add_argument("kernel")
    p.add_argument("--weights")
    p.add_argument("--blocks", action="store_true")
    p.add_argument("--samples-per-wave", type=float, default=64*9, help="supersamples a wave shades (64 lanes x WALK)")
    p.add_argument("--dump", help="write the blocks (label, counts, lines, targets) as JSON")
    args = p.parse_args()
    name, blocks = parse(args.listing, args.kernel)
    if not blocks:
        sys.exit(f"no kernel matching '{args.kernel}' in {args.listing}")
    weights = json.load(open(args.weights)) if args.weights else None
    report(name, blocks, weights, args.samples_per_wave, args.blocks)
    if args.dump:
        json.dump([{"label": b.label, "counts": dict(b.counts), "sgpr_forms": b.sgpr_forms, "lines": dict(b.lines), "targets": b.targets,
                    "instructions": [f"{m} {o}" for m, o in b.instructions]} for b in blocks], open(args.dump, "w"), indent=0)


if __name__ == "__main__":
    main()
