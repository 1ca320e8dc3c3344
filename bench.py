#!/usr/bin/env python3
"""
bench.py — frames/s of the music-visualizer export path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N … bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric config, `configs[2]`): Visualizer scene, 3840x2160, 2x SSAA, subsample 2, 60 fps,
synthetic 44.1 kHz stereo sine sweep, synthetic 1920x1080 background; inputs resident in HBM before timing.
One STEP = one batch of 300 frames (5 s of video; --frames-per-step) through the whole hot path: STFT of the batch's frames →
filterbank → DynamicNumber scan → waveform/loudness → per-frame column/row tables → the batch's fused fragment+SSAA-resolve
frames written as RGB8 into a device frame buffer (two alternating buffers of 7.5 GB each).

N > 1 (one rank per GPU, weak scaling): rank r renders ITS OWN contiguous range of the clip — frames
[r*(W+K)*F, (r+1)*(W+K)*F) — after replaying the DynamicNumber recurrences of everything before it (untimed set-up, as
in the sharded export's device mode, shaderflow_amd/parallel.py), and every finished step is sent to rank 0 over
RCCL/xGMI (grouped point-to-point = the gather of the north star), the transfer of step i overlapping the render of
step i+1. `value` = frames of all ranks / max-over-ranks time: frames resident in rank 0's HBM.

Prints ONE JSON line (rank 0): value = frames/s of the whole job, plus
  roofline     — the dominant kernel (named by the library: sfx_last_kernel). It is bound by VALU issue and LDS bandwidth, not
                 by HBM, so `bound` is "valu": achieved = instruction lanes per second from the launch time measured HERE (HIP
                 events on the launch stream inside the timed region) x the instructions per supersample rocprofv3 counts IN THIS
                 RUN (five short children of this script under `rocprofv3 --kernel-trace --pmc`, one per pass: the VALU class
                 counters, GRBM_GUI_ACTIVE, FETCH_SIZE, WRITE_SIZE, the LDS and wave-state counters; `--no-live-counters` skips
                 them). Where the profiler is missing or a pass fails the tracked profile of THIS build stands in (profiles/*.json,
                 checked against the library's source fingerprint — stale counters are dropped, loudly);
                 `roofline.counters_from` says which it was. The HBM view the contract asks for sits beside it under `hbm`: algorithmic bytes over
                 the same launch time, and `traffic` = measured FETCH_SIZE + WRITE_SIZE per launch from the same profile.
  cpu_baseline — the oracle (kind "port": plain-C restatement of the reference path) on the host cores of this box, all cores
                 and one thread, over bands of three frames of the same workload; rank 0 at N = 1 only, AFTER the GPU legs.
                 `llvmpipe_container`: the reference itself (numpy FFT + its GLSL on Mesa llvmpipe), measured in the build container.
  export_host  — the same frames through a real export to /dev/null including the read-out to host memory (N = 1: pinned ring
                 + writer thread; N > 1: every rank over its own PCIe link into the shared-memory ring, "host" mode).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

B_ALG_PER_FRAME = {  # iScreen write + resolve read + iFinal write + read-out read, bytes
    (3840, 2160, 2): 315.2e6, (1920, 1080, 1): 29.0e6, (256, 256, 1): 0.92e6, (7680, 4320, 4): 4445.8e6,
}
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANE_OPS = 256*4*32*2.4e9    # 256 CU x 4 SIMD32 x 2.4 GHz = 78.6e12 lane-instructions/s (157.3 TFLOP/s FMA)
PROFILE = ROOT/"profiles"/"r06_bench_c3.json"      # written by tools/profile_bench.sh → tools/summarize_profile.py
# The north star's CPU baseline measured with THE REFERENCE ITSELF (its own Python + numpy FFT + its GLSL on Mesa llvmpipe), in the
# build container — it cannot travel to the GPU box (tools/measure_reference_cpu.py, profiles/r03_reference_llvmpipe.txt). Static.
REFERENCE_LLVMPIPE = {"value": 0.477, "unit": "frames/s", "cores": 8, "kind": "reference",
                      "sample": "8 frames of the reference's Visualizer scene, 3840x2160 2xSSAA, scene.main() on Mesa llvmpipe 23.2.1 (LLVM 15, 256-bit), "
                                "8 logical cores of the build container, 16.8 s; C2 (1920x1080, no SSAA): 3.70 frames/s",
                      "where": "build container, not this box (static block)"}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=8)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--width", type=int, default=3840)
    p.add_argument("--height", type=int, default=2160)
    p.add_argument("--ssaa", type=int, default=2)
    p.add_argument("--frames-per-step", type=int, default=300,
                   help="frames of one step = one batch through the hot path = ONE launch of the dominant kernel (grid.z); 300 = 5 s of video, "
                        "so that the driver's 20 timed steps are 2 s of GPU time (round 3's 60-frame steps: 0.41 s, too short for its utilisation sampler)")
    p.add_argument("--seconds", type=float, default=60.0, help="length of the synthetic clip (grown when the ranks need more frames)")
    p.add_argument("--scene", choices=("visualizer", "bars", "waveform", "basic"), default="visualizer",
                   help="visualizer = the metric's scene; bars = MusicBars, waveform = Waveform, basic = Basic (default.glsl): light fragments")
    p.add_argument("--camera-zoom", type=float, default=1.0,
                   help="tier analysis of the light fragments (tools/experiments/basic_tiers.sh): the camera's zoom; the metric's runs leave it at 1")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-export", action="store_true", help="skip the host-inclusive export measurement")
    p.add_argument("--no-live-counters", action="store_true",
                   help="do not collect the dominant kernel's instruction counters in this run (two short rocprofv3 children of this script); "
                        "the roofline then rests on the tracked profile alone")
    p.add_argument("--cpu-seconds", type=float, default=18.0, help="budget of the CPU baseline (all cores + one thread)")
    return p.parse_args()


def host_cores() -> tuple[int, str]:
    """Cores this process may really use: the CPUs it may be scheduled on, capped by the cgroup's CPU quota. The pool's GPU boxes show
    256 logical CPUs and grant 16 CPUs' worth of time (cpu.max = 1600000 100000): round 3's "0.064 frames/s on 256 threads" was 16
    cores throttled by 256 runnable threads (tools/experiments/oracle_scaling.py: 17.7x one thread at 16 threads, 12x at 256)."""
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        limit, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if limit != "max":
            quota = max(1, int(float(limit)/float(period) + 0.5))
    except (OSError, ValueError):
        try:
            limit, period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if limit > 0:
                quota = max(1, int(limit/period + 0.5))
        except (OSError, ValueError):
            pass
    cores = min(visible, quota) if quota else visible
    return cores, f"{visible} logical CPUs visible, cgroup quota {quota if quota else 'none'}"


def cpu_baseline(args, pcm, background) -> dict:
    """Oracle on the host cores: bands of THREE frames of the same workload on all cores, one band on a single thread."""
    import numpy as np

    from oracle import binding as O
    w, h, s = args.width, args.height, args.ssaa
    threads, quota = host_cores()
    planar = np.ascontiguousarray(pcm.T)
    fmin, fmax, bins = O.from_notes(O.lib().sfo_note_of_frequency(20.0, 440.0), O.lib().sfo_note_of_frequency(14000.0, 440.0), True)
    indptr, indices, data = O.filterbank(0, 0, fmin, fmax, bins, 12, 44100)
    bg = O.make_texture(np.flipud(background), "linear", True, True)

    def frame_inputs(k: int):
        tell = 735*k
        t0 = time.perf_counter()
        column = O.csr_dot(indptr, indices, data, O.fft_power(planar, tell))
        row = O.waveform_row(planar, tell, 735, 180)
        _, std = O.volume_std(planar, tell, 4410)
        audio_s = time.perf_counter() - t0
        u = O.default_uniforms(w, h, iTime=k/60.0, iTau=(k/60.0)/args.seconds, iDuration=args.seconds, iSSAA=float(s), iAudioVolume=0.97, iAudioSTD=float(std),
                               iSpectrogramBins=bins, iSpectrogramLength=1, iWaveformLength=180)
        tex = {"background": bg, "iSpectrogram": O.make_texture(column.reshape(bins, 1, 2), "nearest", True, False),
               "iWaveform": O.make_texture(row.reshape(1, 180, 2), "linear", False, False)}
        return u, tex, audio_s

    def band(u, tex, y0: int, rows: int, n_threads: int) -> float:
        t = time.perf_counter()
        screen = O.render("visualizer", u, tex, w*s, h*s, rows=(y0*s, (y0 + rows)*s), threads=n_threads)
        O.resolve(screen, w, h, 2, rows=(y0, y0 + rows), threads=n_threads)
        return time.perf_counter() - t

    u, tex, audio_s = frame_inputs(600)
    # every thread gets whole rows, and at least two of them: round 3 measured bands of threads/4 … threads rows, i.e. most of the 256
    # threads of the GPU box idle or starting up — 0.064 frames/s "on 256 threads" was a statement about the sample, not the cores
    probe_rows = min(h, 2*threads)
    probe = band(u, tex, (h - probe_rows)//2, probe_rows, threads)  # calibrate: seconds per output row on all cores
    per_row = probe/probe_rows
    budget_all, budget_one = 0.7*args.cpu_seconds, 0.3*args.cpu_seconds
    rows_all = int(min(h, max(2*threads, budget_all/3.0/per_row)))
    rows_all -= rows_all % threads if rows_all >= threads else 0
    frames = (300, 1500, 2700)                                     # three frames spread over the clip (different audio, zoom, blur radius)
    seconds_all, audio_all = 0.0, 0.0
    for k in frames:
        u, tex, a_s = frame_inputs(k)
        seconds_all += band(u, tex, (h - rows_all)//2, rows_all, threads)
        audio_all += a_s
    frame_s_all = (seconds_all/len(frames))*(h/rows_all) + audio_all/len(frames)
    probe_one = band(u, tex, h//2, 1, 1)                            # one output row on one thread
    rows_one = int(max(1, min(rows_all, budget_one/max(probe_one, 1e-6))))
    seconds_one = band(u, tex, (h - rows_one)//2, rows_one, 1)
    frame_s_one = seconds_one*(h/rows_one) + audio_all/len(frames)
    return {"value": 1.0/frame_s_all, "unit": "frames/s", "cores": threads, "kind": "port", "host": quota,
            "single_thread": {"value": 1.0/frame_s_one, "unit": "frames/s", "cores": 1,
                              "sample": f"{rows_one} of {h} output rows of one frame ({seconds_one:.1f} s), scaled to a whole frame"},
            "sample": f"{rows_all} of {h} output rows of each of {len(frames)} frames {frames} of the {w}x{h} {s}xSSAA visualizer clip "
                      f"({seconds_all:.1f} s on {threads} threads) + the audio tape of those frames ({audio_all*1e3:.1f} ms), scaled to whole frames"}


def profile_counters(kernel: str) -> dict | None:
    """Counters rocprofv3 collected for `kernel` with THIS build (tools/profile_bench.sh); None — with a warning — when the profile
    is missing, names another kernel or was taken from other kernel sources"""
    from shaderflow_amd._native import source_fingerprint
    try:
        record = json.loads(PROFILE.read_text())
    except (OSError, ValueError):
        print(f"bench.py: no profile at {PROFILE}: roofline.instructions/traffic are null", file=sys.stderr)
        return None
    if record.get("source_fingerprint") != source_fingerprint():
        print(f"bench.py: {PROFILE.name} was measured on other kernel sources ({record.get('source_fingerprint')} != {source_fingerprint()}): "
              "its counters are NOT used; re-run tools/profile_bench.sh", file=sys.stderr)
        return None
    for name, entry in record.get("kernels", {}).items():
        if name.replace(" ", "") == kernel.replace(" ", ""):
            line = record.get("bench_under_tracer") or {}
            return dict(entry, frames_per_launch=(line.get("roofline") or {}).get("frames_per_launch"))
    print(f"bench.py: {PROFILE.name} has no counters for '{kernel}' (it holds {list(record.get('kernels', {}))[:4]}…)", file=sys.stderr)
    return None


LIVE_PASSES = (("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT", "SQ_THREAD_CYCLES_VALU"),
               ("GRBM_GUI_ACTIVE", "GRBM_COUNT"),
               ("FETCH_SIZE",), ("WRITE_SIZE",),                     # (each in a pass of its own, as the microarchitecture guide prescribes)
               ("SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"))


def clean_kernel_name(raw: str) -> str:
    return raw.replace("sf::", "").replace("void ", "").split("(")[0].strip()


def live_counters(args, kernel: str | None) -> dict | None:
    """The instruction counters of this configuration's DOMINANT kernel measured IN THIS RUN: one short child of this script per pass under
    `rocprofv3 --kernel-trace --pmc` (counters in runs of their own, the program itself after `--`), this run's frames per launch. The
    dominant kernel is the one the trace gives the most GPU time (C2's last launch is the resolve, its dominant kernel the strip kernel:
    `kernel`, the library's last launch, only breaks ties and names the kernel when the trace is empty). The same structure as
    profile_counters() plus `kernel` and `share` (its part of the traced GPU time); None — with the reason on stderr — when the profiler
    is not there or a pass fails (the tracked profile stands)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        print("bench.py: no rocprofv3 on PATH: the roofline's counters come from the tracked profile", file=sys.stderr)
        return None
    frames = args.frames_per_step
    per_kernel: dict = {}                                           # kernel -> counter -> values
    durations: dict = {}                                            # kernel -> durations (ns) of the GRBM pass' trace
    with tempfile.TemporaryDirectory(prefix="shaderflow_bench_pmc_", dir="/tmp") as scratch:
        for index, names in enumerate(LIVE_PASSES):
            out = Path(scratch)/f"pass{index}"
            command = ["rocprofv3", "--kernel-trace", "--pmc", *names, "-f", "csv", "-d", str(out), "-o", "pmc", "--",
                       sys.executable, str(Path(__file__).resolve()), "--steps", "1", "--warmup", "1", "--frames-per-step", str(frames),
                       "--width", str(args.width), "--height", str(args.height), "--ssaa", str(args.ssaa), "--scene", args.scene,
                       "--camera-zoom", str(args.camera_zoom), "--no-cpu-baseline", "--no-export", "--no-live-counters"]
            try:                                                     # (a session of its own: a pass that overruns is ended with its whole process group)
                child = subprocess.Popen(command, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, TMPDIR="/tmp"),
                                         cwd=str(ROOT), start_new_session=True)
            except OSError as error:
                print(f"bench.py: counter pass {index} did not start ({error}): the tracked profile stands", file=sys.stderr)
                return None
            try:
                _, errors = child.communicate(timeout=90)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(child.pid, signal.SIGKILL)
                child.communicate()
                print(f"bench.py: counter pass {index} took more than 90 s and was ended: the tracked profile stands", file=sys.stderr)
                return None
            found = 0
            for table in out.glob("**/*counter_collection.csv"):
                with open(table) as handle:
                    for row in csv.DictReader(handle):
                        per_kernel.setdefault(clean_kernel_name(row["Kernel_Name"]), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                        found += 1
            if child.returncode != 0 or not found:
                print(f"bench.py: counter pass {index} gave nothing (rc {child.returncode}): the tracked profile stands\n{errors[-400:]}", file=sys.stderr)
                return None
            if "GRBM_GUI_ACTIVE" in names:                          # the pass GRBM_GUI_ACTIVE came from: its own durations give the clock
                for table in out.glob("**/*kernel_trace.csv"):
                    with open(table) as handle:
                        for row in csv.DictReader(handle):
                            durations.setdefault(clean_kernel_name(row["Kernel_Name"]), []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    squeezed = {name.replace(" ", ""): name for name in per_kernel}
    # the render kernels of this library only (torch's fills, RCCL's and the audio tape's kernels are not the roofline's subject)
    candidates = {name: sum(values) for name, values in durations.items() if name.startswith("k_") and not name.startswith(("k_stft", "k_dft", "k_filterbank", "k_dynamics", "k_waveform", "k_volume", "k_spectrogram"))}
    total = sum(candidates.values())
    dominant = max(candidates, key=candidates.get) if candidates else kernel
    if dominant is None or dominant.replace(" ", "") not in squeezed:
        print(f"bench.py: the counter passes hold nothing for '{dominant}': the tracked profile stands", file=sys.stderr)
        return None
    counters = {name: sum(values)/len(values) for name, values in per_kernel[squeezed[dominant.replace(" ", "")]].items()}
    spans = durations.get(dominant) or []
    # launches of the dominant kernel per traced step: the counters are per LAUNCH, a step of this run may take several (none today)
    return {"kernel": dominant, "counters": counters, "duration": {"average_ns": sum(spans)/len(spans)} if spans else None, "frames_per_launch": frames,
            "share": round(candidates[dominant]/total, 4) if total and dominant in candidates else None,
            "others": {name: round(value/total, 4) for name, value in sorted(candidates.items(), key=lambda item: -item[1])[1:4]} if total else None}


# SIMD cycles a wave64 VALU instruction of each class occupies, as measured on this chip (tools/ubench_valu.hip,
# profiles/r02_ubench_valu.txt: add/mul/fma f32 2.3-2.6 cycles at the nominal clock = 2 at the clock the loop sustained; conversions,
# integer multiply-adds, shifts, min/max/perm, anything with an SGPR operand 4.1-4.4; v_rcp/v_sqrt/v_exp/v_log 8.2-8.7)
CLASS_CYCLES = {"SQ_INSTS_VALU_ADD_F32": 2, "SQ_INSTS_VALU_MUL_F32": 2, "SQ_INSTS_VALU_FMA_F32": 2, "SQ_INSTS_VALU_INT32": 4, "SQ_INSTS_VALU_CVT": 4,
                "SQ_INSTS_VALU_TRANS_F32": 8, "SQ_INSTS_VALU_ADD_F16": 4, "SQ_INSTS_VALU_FMA_F16": 4}


CENSUS = ROOT/"profiles"/"r06_strip_isa_census.json"      # tools/strip_census.py --json: what the uncounted VALU instructions ARE


def census_prices(kernel: str) -> dict | None:
    """The ISA census of the strip kernel (profiles/r05_strip_isa_census.txt): issue cycles of the instructions no hardware class counter
    covers, and how many full-rate forms take a scalar source (+2 cycles each). Only for the kernel and the sources it was counted on."""
    import hashlib
    try:
        census = json.loads(CENSUS.read_text())
    except (OSError, ValueError):
        return None
    if census.get("kernel", "").replace(" ", "") != kernel.replace(" ", ""):
        return None
    digest = hashlib.sha256()
    csrc = ROOT/"shaderflow_amd"/"csrc"
    for name in ("visualizer_fast.hpp", "visualizer_kernels.hpp", "render_kernels.hpp", "fragments.hpp", "glsl.hpp", "sfmath.hpp", "launch_visualizer_strip.hip", "Makefile"):
        digest.update(name.encode()); digest.update((csrc/name).read_bytes())
    if digest.hexdigest()[:16] != census.get("strip_sources_fingerprint"):
        print(f"bench.py: {CENSUS.name} was counted on other kernel sources: issue_model keeps its 2-or-4 band; re-run tools/strip_census.py", file=sys.stderr)
        return None
    return census


def issue_model(cs: dict, launch_ns: float | None, census: dict | None = None) -> dict | None:
    """VALU issue cycles the profiled launch NEEDED (rocprofv3's per-class instruction counters x the measured cycles per class; the
    instructions no class counter covers — moves, selects, compares, min/max, bit operations — priced at 2 cycles for `frac_low` and at
    4 for `frac`) over the SIMD cycles it HAD (GRBM_GUI_ACTIVE counts every XCD: / 8 = cycles of the launch at the clock the chip
    actually ran, x 256 CUs x 4 SIMDs). The honest roofline of this kernel: how busy its issue ports are."""
    if not cs.get("SQ_INSTS_VALU") or not cs.get("GRBM_GUI_ACTIVE") or "SQ_INSTS_VALU_FMA_F32" not in cs:
        return None
    classes = {name: cs.get(name, 0.0) for name in CLASS_CYCLES}
    counted = sum(classes.values())
    other = max(0.0, cs["SQ_INSTS_VALU"] - counted)
    priced = sum(classes[name]*cycles for name, cycles in CLASS_CYCLES.items())
    cycles = cs["GRBM_GUI_ACTIVE"]/8.0
    had = cycles*256*4
    by_census = None
    if census:
        # one number instead of the band: the uncounted instructions at the price of what the ISA census found them to
        # be, plus the two extra cycles every full-rate form with a scalar source takes (v_mov_b32 v, s: 4.2 cycles, r05_ubench_valu_sgpr.txt)
        # … and the two extra cycles of every packed-f32 form: the class counters count a v_pk_fma_f32 once, as an fma (r05_ubench_pk_f32.txt)
        by_census = round((priced + census["other_cycles_per_instruction"]*other
                           + 2.0*census["sgpr_source_full_rate_forms_per_valu_instruction"]*cs["SQ_INSTS_VALU"]
                           + 2.0*census.get("packed_f32_forms_per_valu_instruction", 0.0)*cs["SQ_INSTS_VALU"])/had, 4)
    return {"frac": by_census if by_census is not None else round((priced + 4*other)/had, 4), "frac_census": by_census,
            "frac_other_at_4": round((priced + 4*other)/had, 4), "frac_low": round((priced + 2*other)/had, 4),
            "census": ({"other_cycles_per_instruction": census["other_cycles_per_instruction"],
                        "sgpr_source_full_rate_forms_per_valu_instruction": census["sgpr_source_full_rate_forms_per_valu_instruction"],
                        "packed_f32_forms_per_valu_instruction": census.get("packed_f32_forms_per_valu_instruction", 0.0),
                        "from": str(CENSUS.relative_to(ROOT))} if census else None),
            "simd_cycles_available": had, "effective_clock_GHz": round(cycles/launch_ns, 3) if launch_ns else None,
            "instructions": {"add_f32": classes["SQ_INSTS_VALU_ADD_F32"], "mul_f32": classes["SQ_INSTS_VALU_MUL_F32"], "fma_f32": classes["SQ_INSTS_VALU_FMA_F32"],
                             "int32": classes["SQ_INSTS_VALU_INT32"], "cvt": classes["SQ_INSTS_VALU_CVT"], "trans_f32": classes["SQ_INSTS_VALU_TRANS_F32"],
                             "other": other, "all": cs["SQ_INSTS_VALU"]},
            "wave_cycles": {"active_inst_any": cs.get("SQ_ACTIVE_INST_ANY"), "wait_inst_any": cs.get("SQ_WAIT_INST_ANY"), "wait_any": cs.get("SQ_WAIT_ANY"),
                            "all": cs.get("SQ_WAVE_CYCLES")},
            "lane_utilisation": round(cs["SQ_THREAD_CYCLES_VALU"]/(cs["SQ_INSTS_VALU"]*64.0), 4) if cs.get("SQ_THREAD_CYCLES_VALU") else None}


def refuse(sentence: str) -> "NoReturn":
    """One sentence on stderr and a non-zero exit: a line that says `n_gpus: 1` when N were asked for must never be printed"""
    print(f"bench.py: {sentence}", file=sys.stderr, flush=True)
    raise SystemExit(2)


def launcher_command(gpus: int, argv: list[str], port: int) -> list[str]:
    """`python bench.py --gpus N` with no launcher around it: the command of the N ranks, exactly what the driver itself uses for N > 1"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), str(Path(__file__).resolve()), *argv]


def spawn_ranks(args, argv: list[str]) -> int:
    """--gpus N > 1 and no WORLD_SIZE: this process BECOMES the launcher's parent. Nothing here has imported torch or the HIP library
    (no GPU call: the ranks are a CHILD process, never an exec of a process that touched the GPU); rank 0's JSON line is relayed as
    this process' last stdout line, everything else the ranks print goes to stderr, and the child's exit code is this one's."""
    import socket
    import subprocess
    with socket.socket() as probe:                                  # a free port for the rendezvous (the driver passes its own when IT launches)
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    command = launcher_command(args.gpus, argv, port)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), SHADERFLOW_BENCH_SPAWNED="1")
    print(f"bench.py: --gpus {args.gpus} without a launcher: starting {' '.join(command[1:8])} …", file=sys.stderr, flush=True)
    # (a session of its own, and this process' termination is the ranks': a driver that ends the bench on its clock must not leave N
    # processes holding the GPUs)
    import signal
    child = subprocess.Popen(command, stdout=subprocess.PIPE, text=True, env=env, cwd=str(ROOT), start_new_session=True)

    def end_ranks(signum=None, frame=None):
        if child.poll() is None:
            try:
                os.killpg(child.pid, signal.SIGTERM)
                child.wait(timeout=10)
            except (OSError, subprocess.TimeoutExpired):
                try:
                    os.killpg(child.pid, signal.SIGKILL)
                except OSError:
                    pass
        if signum is not None:
            raise SystemExit(128 + signum)
    for name in ("SIGTERM", "SIGINT", "SIGHUP"):
        signal.signal(getattr(signal, name), end_ranks)
    import atexit
    atexit.register(end_ranks)
    line = None
    for text in child.stdout:
        candidate = text.strip()
        record = None
        if candidate.startswith("{") and candidate.endswith("}"):
            try:
                record = json.loads(candidate)
            except ValueError:
                record = None
        if isinstance(record, dict) and "metric" in record and "n_gpus" in record:
            line = candidate
        else:
            sys.stderr.write(text)
    rc = child.wait()
    if rc != 0:
        print(f"bench.py: the {args.gpus} ranks ended with exit code {rc}: no line", file=sys.stderr, flush=True)
        return rc
    if line is None:
        print(f"bench.py: the {args.gpus} ranks ended without a JSON line", file=sys.stderr, flush=True)
        return 3
    record = json.loads(line)
    if record.get("n_gpus") != args.gpus or record.get("rccl_ranks") != args.gpus:
        print(f"bench.py: asked for {args.gpus} GPUs, the line says n_gpus {record.get('n_gpus')} / rccl_ranks {record.get('rccl_ranks')}: refused", file=sys.stderr, flush=True)
        return 4
    sys.stderr.flush()
    print(line, flush=True)
    return 0


def main() -> None:
    args = parse_args()
    if args.gpus < 1:
        refuse(f"--gpus {args.gpus}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # (before torch, before the HIP library, before any GPU call)
        raise SystemExit(spawn_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # either way the line would not be what was asked for: N ranks under --gpus 1, or ONE rank under --gpus N (a launcher that set
        # WORLD_SIZE=1) — the latter printed `n_gpus: 1` silently until round 5
        refuse(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")

    # the host driver of this pool supports dmabuf IPC only: without this RCCL's cross-process buffers fail (hipIpcGetMemHandle)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch                                                  # before the HIP library: one HIP runtime per process

    backend = os.environ.get("SHADERFLOW_DIST_BACKEND", "nccl")
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        refuse(f"--gpus {world} under RCCL needs {world} visible GPUs, this node shows {torch.cuda.device_count()} (one rank per GPU; several ranks on one device is the gloo test's set-up)")
    local_rank %= max(1, torch.cuda.device_count())               # more ranks than devices only happens in the gloo test
    torch.cuda.set_device(local_rank)
    distributed = world > 1 or os.environ.get("SHADERFLOW_FORCE_DIST") == "1"     # the env var exercises the RCCL path on one GPU
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") is the backend; SHADERFLOW_DIST_BACKEND=gloo lets tests run several ranks on ONE device (RCCL refuses that)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from examples.scenes import Basic, MusicBars, Visualizer, Waveform, make
    from shaderflow_amd import _native as N
    from shaderflow_amd import synth
    from shaderflow_amd.message import ShaderMessage
    from shaderflow_amd.tape import FrameTape

    w, h, s, fpb = args.width, args.height, args.ssaa, args.frames_per_step
    total_steps = args.warmup + args.steps
    frames_per_rank = total_steps*fpb
    seconds = max(args.seconds, world*frames_per_rank/60.0)       # every rank gets its own contiguous range of the clip
    pcm = synth.sweep_clip(seconds, 44100)
    background = synth.background_image(1920, 1080, seed=0)
    scene_class = {"visualizer": Visualizer, "bars": MusicBars, "waveform": Waveform, "basic": Basic}[args.scene]

    # Everything of this rank runs on ONE stream, which is also torch's current stream: RCCL orders a send after the work already
    # queued on the current stream, and a later render waits for the transfer it must not overtake. (torch's default stream has
    # the handle 0, for which the context would create a stream of its own that nothing orders with — so a real one.)
    render_stream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(render_stream)
    context = N.Context(local_rank, render_stream.cuda_stream)
    count_fallbacks = os.environ.get("SHADERFLOW_BENCH_TILE_MISSES") == "1"     # diagnostics: blocks / waves that left their kernel's fast tier (sfx_ctx_tile_misses)
    if count_fallbacks:
        context.tile_misses()                                     # allocates and zeroes the device counter

    def build_scene(prepared: bool = True):
        scene = make(scene_class, audio=(None if args.scene == "basic" else (pcm, 44100)), background=(background if args.scene == "visualizer" else None), context=context)
        if not prepared:
            return scene                                          # scene.main() does the rest itself
        scene.initialize()
        scene.exporting = scene.freewheel = scene.headless = True
        scene.realtime = False
        scene.fps, scene.subsample, scene.time = 60.0, 2, 0.0
        scene.relay(ShaderMessage.Shader.Compile)
        scene.resize(width=w, height=h)
        for module in scene.modules:
            module.setup()
        scene.set_duration(seconds)
        scene.ssaa = s
        if args.camera_zoom != 1.0:
            scene.camera.zoom.set(args.camera_zoom)
        return scene

    scene = build_scene()
    first_frame = rank*frames_per_rank                              # this rank's contiguous range: [first_frame, first_frame + frames_per_rank)
    tape = FrameTape(scene, batch=fpb).prepare(first_frame + frames_per_rank)
    tape.bind_static_uniforms()

    frame_bytes = w*h*3
    yuv_bytes = w*h*3//2
    # zeros, not empty: the first touch of fresh device memory is paid here, outside the timed region
    buffers = [torch.zeros(fpb*frame_bytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
    # N > 1: a step's frames are rendered and sent in `parts` pieces, so that only the last piece's transfer is exposed at the end
    # of the timed region (the transfer of a piece overlaps the render of the next one); N = 1 renders the batch at once
    parts = next(p for p in (4, 3, 2, 1) if fpb % p == 0) if distributed else 1
    piece = fpb//parts
    staged = distributed and dist.get_backend() == "gloo"
    loopback = distributed and world == 1 and dist.get_backend() == "nccl"

    # ---- the gather (N > 1): every finished piece → rank 0's HBM. Two transports over the same buffers, measured as LEGS of this run:
    #   "p2p"   grouped point-to-point calls of the process group's backend (RCCL over xGMI: the north star's gather; gloo in tests)
    #   "sdma"  peer copies through IPC windows on SDMA engines NAMED through HSA (shaderflow_amd/parallel.py SdmaTransfer, sfx_peer_*)
    # and two payloads: the rgb24 frames (the reference's byte stream: `value`) and — converted on the rank that rendered them —
    # planar yuv420p, half the bytes per link (`yuv420p` beside it). SHADERFLOW_SHARD=device|device-sdma pins the transport.
    from shaderflow_amd.parallel import DeviceArray
    os.environ.setdefault("SHADERFLOW_COPY_TIMEOUT", "30")          # an engine copy that never completes is reported after 30 s (capi_readout.hip EngineLanes::finish), not waited for
    pinned = os.environ.get("SHADERFLOW_SHARD", "").strip().lower()
    transports = [] if not distributed else (["sdma"] if pinned == "device-sdma" else (["p2p"] if pinned == "device" else ["p2p", "sdma"]))
    received = raw_received = windows = None
    planar_buffers = None
    if distributed:
        planar_buffers = [torch.zeros(fpb*yuv_bytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
        if rank == 0:
            # rank 0 keeps the last two steps of every other rank (a real export hands them to the sink: parallel.contiguous_device_export);
            # raw allocations, so that the SAME memory serves RCCL's receives (as torch views) and the peers' IPC windows
            raw_received = [[context.alloc(fpb*frame_bytes) for _ in range(2)] for _ in range(world)]
            received = [[torch.as_tensor(DeviceArray(pointer, fpb*frame_bytes), device=torch.device("cuda", local_rank)) for pointer in pair] for pair in raw_received]
            for pair in received:
                for tensor in pair:
                    tensor.zero_()
            torch.cuda.synchronize()
            if staged:                                                # gloo carries host tensors only: the receives land in host staging (tests)
                received = [[torch.zeros(fpb*frame_bytes, dtype=torch.uint8, device="cpu") for _ in range(2)] for _ in range(world)]
        sdma_dropped = None
        if "sdma" in transports:
            # PREFLIGHT, collectively: export / open the windows and move one small piece through them. The peer copies have run on ONE
            # GPU only (a loopback window); if anything in them does not hold between two real GPUs, every rank drops the transport
            # together — the run keeps its RCCL legs and says why — instead of one rank raising inside a timed leg while the others wait.
            problem = ""
            handles = [None]
            try:
                if rank == 0:
                    handles = [[[context.peer_export(pointer) for pointer in pair] for pair in raw_received]]
            except Exception as error:                               # noqa: BLE001 — whatever it is, the other ranks must hear of it
                problem = f"rank 0 export: {error}"
            dist.broadcast_object_list(handles, src=0)
            try:
                if not problem and handles[0] is not None:
                    if rank:
                        windows = [context.peer_open(handle) for handle in handles[0][rank]]
                    elif loopback:
                        windows = list(raw_received[0])             # ONE rank: the copies go to this rank's own receive buffers (no IPC mapping of one's own allocation)
                    if os.environ.get("SHADERFLOW_BENCH_INJECT") == "sdma-preflight" and rank == world - 1:
                        raise RuntimeError("injected: this rank's peer copies do not work")     # tests: the run must go on without the transport
                    if windows and (rank or loopback):
                        probe = torch.full((4096,), 0xA5, dtype=torch.uint8, device="cuda")
                        torch.cuda.current_stream().synchronize()
                        context.peer_copy(windows[0], probe.data_ptr(), probe.numel(), lane=0)
                        context.peer_flush()
                elif rank:
                    problem = "no window handles from rank 0"
            except Exception as error:                               # noqa: BLE001
                problem = f"rank {rank}: {error}"
            problems = [None]*world
            dist.all_gather_object(problems, problem)
            if any(problems):
                sdma_dropped = "; ".join(text for text in problems if text)
                transports = [t for t in transports if t != "sdma"] or ["p2p"]
                if rank == 0:
                    print(f"bench.py: SDMA peer copies dropped from this run ({sdma_dropped})", file=sys.stderr)

    class Gather:
        """One transport x one payload: send(index, q, view) queues piece q of step `index`; drain() bounds what is in flight"""
        def __init__(self, transport: str, payload_bytes: int):
            self.transport, self.payload = transport, payload_bytes
            # one entry per TRANSFER (a piece of a step): the works its grouped call returned — RCCL coalesces a batch into ONE work,
            # gloo returns one per operation, so the count of works says nothing about the count of transfers
            self.in_flight: list[list] = []

        def send(self, index: int, q: int, view) -> None:
            lo, hi = q*piece*self.payload, (q + 1)*piece*self.payload
            if self.transport == "sdma":
                if rank or loopback:
                    context.peer_copy(windows[index % 2] + lo, view.data_ptr(), hi - lo, lane=(index % 2)*parts + q)
                return
            if loopback:
                # ONE rank over RCCL (SHADERFLOW_FORCE_DIST=1): the piece is sent to this rank itself — the same grouped point-to-point
                # call, RCCL's own kernel on the render stream, the same ordering against the next render — so that everything the
                # first multi-GPU run depends on has executed on a single GPU before (tests/test_gpu_rccl.py)
                ops = [dist.P2POp(dist.irecv, received[0][index % 2][lo:hi], 0), dist.P2POp(dist.isend, view, 0)]
            elif rank == 0:
                ops = [dist.P2POp(dist.irecv, received[source][index % 2][lo:hi], source) for source in range(1, world)]
            else:
                ops = [dist.P2POp(dist.isend, view.cpu() if staged else view, 0)]
            if ops:
                self.in_flight.append(list(dist.batch_isend_irecv(ops)))

        def fence(self, index: int, q: int) -> None:
            if self.transport == "sdma" and (rank or loopback):
                context.peer_fence((index % 2)*parts + q)          # the copy that last read this piece of this buffer (two steps ago) has left it

        def drain(self, keep_transfers: int = 0) -> None:
            while len(self.in_flight) > keep_transfers:
                for work in self.in_flight.pop(0):
                    work.wait()

        def flush(self) -> None:
            self.drain()
            if self.transport == "sdma" and (rank or loopback):
                context.peer_flush()                              # this rank's frames have landed in rank 0's HBM

    def barrier(gather=None):
        if gather is not None:
            gather.flush()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def step(index: int, timed_slot, gather, planar: bool):
        first = first_frame + index*fpb
        target = buffers[index % 2]
        tape.build(first, fpb)
        if timed_slot is not None:
            context.event_record(2*timed_slot)
        for q in range(parts):
            if gather is not None:
                gather.drain(keep_transfers=2*parts - 1)          # the transfer that last used this piece of this buffer (two steps ago) is done
                gather.fence(index, q)
            view = target[q*piece*frame_bytes:(q + 1)*piece*frame_bytes]
            tape.render(piece, view.data_ptr(), first_slot=q*piece)
            if planar:
                out = planar_buffers[index % 2][q*piece*yuv_bytes:(q + 1)*piece*yuv_bytes]
                context.rgb_to_yuv420(view.data_ptr(), out.data_ptr(), w, h, piece)
                view = out
            if gather is not None:
                gather.send(index, q, view)
        if timed_slot is not None:
            context.event_record(2*timed_slot + 1)

    # Set-up, not a measured or counted step: first launches load code objects, commit scratch buffers and bring the GPU out of its
    # idle power state
    started = time.perf_counter()
    launches = 0
    while launches < 2 or time.perf_counter() - started < 0.3:
        tape.build(0, fpb)
        tape.render(fpb, buffers[launches % 2].data_ptr())
        torch.cuda.synchronize()
        launches += 1

    def run_leg(transport, planar: bool) -> dict:
        """W untimed + K timed steps of this rank's range through one transport and payload; the max-over-ranks time of the K steps"""
        gather = Gather(transport, yuv_bytes if planar else frame_bytes) if transport else None
        # the recurrences of every frame before this rank's range are replayed (audio kernels only, no render): every leg starts from
        # the same tape state
        N.check(N.lib().sfx_tape_reset(tape.handle))
        for f in range(0, first_frame, fpb):
            tape.build(f, min(fpb, first_frame - f))
        if gather is not None:
            gather.send(0, 0, (planar_buffers if planar else buffers)[0][:piece*gather.payload])   # the communicator opens its peer-to-peer channels on first use
            gather.flush()
        for i in range(args.warmup):
            step(i, None, gather, planar)
        barrier(gather)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i, i if i < 32 else None, gather, planar)
        barrier(gather)
        elapsed = time.perf_counter() - t0
        intact = None
        if loopback:                                               # what RCCL / the copy engines delivered is what was rendered (last step, both on this device)
            last = args.warmup + args.steps - 1
            sent = (planar_buffers if planar else buffers)[last % 2]
            intact = bool(torch.equal(received[0][last % 2][:sent.numel()], sent))
        if distributed:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if staged else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        events = [context.event_elapsed_ms(2*i, 2*i + 1) for i in range(min(args.steps, 32))]
        leg = {"transport": transport, "payload": "yuv420p" if planar else "rgb24", "elapsed_s": elapsed,
               "value": round(world*args.steps*fpb/elapsed, 2) if elapsed > 0 else None, "events_ms": events, "loopback_intact": intact}
        if distributed:
            payload = yuv_bytes if planar else frame_bytes
            mine = {"rank": rank, "render_ms_per_step": round(float(np.mean(events)), 3) if events else None,
                    "render_frames_per_s": round(fpb/(float(np.mean(events))/1e3), 1) if events else None,
                    "sent_GB_per_s": (round(args.steps*fpb*payload/elapsed/1e9, 2) if (rank or loopback) else 0.0),
                    "peer_copies": context.peer_route() if transport == "sdma" else None}
            leg["per_rank"] = [None]*world
            dist.all_gather_object(leg["per_rank"], mine)
        return leg

    legs = []
    if not distributed:
        legs.append(run_leg(None, False))
    else:
        for transport in transports:
            legs.append(run_leg(transport, False))
        for transport in transports:
            legs.append(run_leg(transport, True))
    # the headline: rgb24 frames (the reference's stream) resident in rank 0's HBM through the faster of the measured transports
    chosen = max((leg for leg in legs if leg["payload"] == "rgb24"), key=lambda leg: leg["value"] or 0.0)
    elapsed = chosen["elapsed_s"]
    sdma = chosen["transport"] == "sdma"
    dist_backend = dist.get_backend() if distributed else None
    loopback_intact = chosen["loopback_intact"]
    ranks_seen, per_rank = world, chosen.get("per_rank")
    if distributed:
        # did every rank take part?
        ones = torch.ones(1, dtype=torch.float64, device="cpu" if staged else "cuda")
        dist.all_reduce(ones)
        ranks_seen = int(round(float(ones.item())))
        if windows and rank:
            for window in windows:
                context.peer_close(window)
        dist.barrier()
        if raw_received:
            del received
            torch.cuda.synchronize()
            for pair in raw_received:
                for pointer in pair:
                    context.free(pointer)

    if count_fallbacks:
        print(f"bench.py: tile misses / tier fallbacks counted on the device: {context.tile_misses()} over {(args.warmup + args.steps)*fpb*len(legs)} frames", file=sys.stderr)
    kernel_ms = chosen["events_ms"]
    launch_s = (float(np.mean(kernel_ms))/1e3/parts) if kernel_ms else float("nan")     # one launch = `piece` frames
    kernel = N.lib().sfx_last_kernel().decode()
    frames_total = world*args.steps*fpb
    value = frames_total/elapsed if elapsed > 0 else float("nan")
    tape.release()

    # ---- the same frames through a real export, read-out to host memory included (/dev/null sink) ----
    export = None
    if not args.no_export:
        scene2 = build_scene(prepared=False)
        # the whole clip when the default workload is measured (60 s = 3 600 frames: the north star's export), a bounded piece otherwise
        whole_clip = (w, h, s) == (3840, 2160, 2) and args.steps >= 8
        frames_export = int(seconds*60) if whole_clip else max(fpb, min(world*args.steps*fpb, int(seconds*60)))
        def timed_export(scene_to_run, pixel_format):
            """scene.main() to /dev/null between two barriers; a rank that fails still reaches the second barrier and tells the others
            (an export leg that does not work between real GPUs must cost the line its `export_host`, not the whole line)"""
            barrier()
            started = time.perf_counter()
            problem = ""
            try:
                extra = {"pixel_format": pixel_format} if pixel_format else {}
                scene_to_run.main(width=w, height=h, ssaa=s, fps=60.0, time=frames_export/60.0, output="/dev/null", **extra)
            except Exception as error:                               # noqa: BLE001
                problem = f"rank {rank}: {type(error).__name__}: {error}"
            barrier()
            seconds_taken = time.perf_counter() - started
            problems = [problem]
            if distributed:
                problems = [None]*world
                dist.all_gather_object(problems, problem)
            problems = [text for text in problems if text]
            if problems and rank == 0:
                print(f"bench.py: export leg ({pixel_format or 'rgb24'}) failed: {'; '.join(problems)}", file=sys.stderr)
            return seconds_taken, "; ".join(problems)

        took, failed = timed_export(scene2, None)
        # the same export with the frames converted to planar yuv420p on the device (scene.main(pixel_format="yuv420p"), opt-in:
        # SURVEY §8 f1's optional half): 12.4 MB per frame over PCIe instead of 24.9 — since round 5 in every shard mode too, converted
        # on the rank that rendered the frame
        took_planar, failed_planar = timed_export(build_scene(prepared=False), "yuv420p")
        planar = {"value": round(frames_export/took_planar, 2) if not failed_planar else None, "unit": "frames/s", "frames": frames_export, "error": failed_planar or None,
                  "note": "BT.601 limited range, chroma from the rounded 2x2 mean; not the reference's byte stream (it hands ffmpeg rgb24)"}
        export = {"value": round(frames_export/took, 2) if not failed else None, "unit": "frames/s", "frames": frames_export, "seconds": round(took, 3), "yuv420p": planar,
                  "error": failed or None,
                  "mode": ("pinned ring + writer thread" if world == 1 else f"sharded export, SHADERFLOW_SHARD={os.environ.get('SHADERFLOW_SHARD', 'host')}"),
                  "note": "whole scene.main(): tape schedule, table set-up, render, read-out over PCIe, write to /dev/null"}

    # the CPU baseline LAST: the GPU legs above are what the driver's utilisation sampler should see first
    baseline = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and args.scene == "visualizer":
        baseline = cpu_baseline(args, pcm, background)
        if (w, h, s) == (3840, 2160, 2):
            baseline["llvmpipe_container"] = REFERENCE_LLVMPIPE

    if ranks_seen != args.gpus:
        # the all-reduce over the process group counted another number of ranks than --gpus: whatever the cause, no line
        refuse(f"rank {rank}: the process group counted {ranks_seen} ranks, --gpus {args.gpus} was asked for")
    if rank == 0:
        c3 = (w, h, s) == (3840, 2160, 2) and args.scene == "visualizer"
        b_alg = B_ALG_PER_FRAME.get((w, h, s), float(w*s*h*s*8 + w*h*6))
        hbm_achieved = b_alg*piece/launch_s/1e9
        samples_per_s = (w*s)*(h*s)*piece/launch_s
        counters = profile_counters(kernel) if c3 else None
        counters_from = str(PROFILE.relative_to(ROOT)) if counters else None
        profiled = any(key.startswith(("ROCPROF", "ROCP_")) for key in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")   # this run IS a profiler's child
        dominant_share = other_kernels = None
        if world == 1 and not distributed and not profiled and not args.no_live_counters:
            # EVERY configuration's run measures its own counters (round 5, missing 3): the dominant kernel by traced
            # GPU time; the tracked profile fills in what a pass did not give (C3 only: it is the profile of C3's kernel)
            live = live_counters(args, kernel)
            if live:
                if live["kernel"].replace(" ", "") != kernel.replace(" ", ""):
                    counters = None                                 # (the tracked profile names the library's last launch)
                kernel = live["kernel"]
                dominant_share, other_kernels = live["share"], live["others"]
                tracked = counters if (counters and counters.get("frames_per_launch") == live["frames_per_launch"]) else None   # per-launch counters of another launch size do not mix
                merged = dict(tracked["counters"]) if tracked else {}
                merged.update(live["counters"])
                counters = {"counters": merged, "duration": live["duration"] or (tracked or {}).get("duration"), "frames_per_launch": live["frames_per_launch"]}
                counters_from = ("live: " + ", ".join(name for names in LIVE_PASSES for name in names) + " (rocprofv3 children of this run)"
                                 + (f"; the rest: {counters_from}" if tracked else ""))
        per_sample = traffic = lds_busy = issue = None
        if counters:
            cs = counters["counters"]
            if cs.get("SQ_INSTS_VALU"):                            # wave-instructions x 64 lanes over the supersamples of the profiled launch
                profiled_frames = counters.get("frames_per_launch") or fpb            # (a lane of the strip kernel shades several supersamples)
                per_sample = cs["SQ_INSTS_VALU"]*64.0/((w*s)*(h*s)*profiled_frames)
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:        # KiB per profiled launch → bytes per launch of `piece` frames
                profiled_frames = counters.get("frames_per_launch") or fpb
                traffic = (cs["FETCH_SIZE"] + cs["WRITE_SIZE"])*1024.0*piece/profiled_frames
            if cs.get("GRBM_GUI_ACTIVE") and cs.get("SQ_LDS_IDX_ACTIVE"):
                lds_busy = cs["SQ_LDS_IDX_ACTIVE"]/(cs["GRBM_GUI_ACTIVE"]/8.0*256.0)     # LDS-array cycles / (cycles x CUs); GRBM counts per XCD
            issue = issue_model(cs, (counters.get("duration") or {}).get("average_ns"), census_prices(kernel))
        lane_ops = samples_per_s*per_sample if per_sample else None
        result = {
            "metric": "frames/sec at 4K 2xSSAA music-visualizer" if c3 else f"frames/sec {args.scene} {w}x{h} {s}xSSAA",
            "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed/max(1, args.steps)*1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{scene_class.__name__} scene {w}x{h} {s}xSSAA subsample 2, 60 fps, {seconds:.0f} s synthetic stereo sine sweep @44.1 kHz"
                                   + (f" (BASELINE's 60 s clip grown so that each of the {world}x({args.warmup}+{args.steps}) steps renders {fpb} NEW frames: same per-frame work)" if seconds > args.seconds else "")
                                   + f", 1920x1080 synthetic background; step = {fpb} frames (STFT + filterbank + dynamics tape, column/row tables, fused fragment+resolve)",
                       "frames_per_step": fpb, "global_frames_per_step": fpb*world,
                       "parallelism": f"contiguous frame range per rank x{world}" + (f", every step sent to rank 0 {'as SDMA peer copies (IPC windows)' if sdma else 'over ' + dist.get_backend()} in {parts} pieces" if distributed else ""),
                       "ranks": world, "filterbank": "mfma" if tape.use_mfma else "csr"},
            "rccl_ranks": ranks_seen,
            "realtime_factor": round(value/60.0, 2),
            "roofline": {"bound": "valu", "kernel": kernel, "kernel_share_of_gpu_time": dominant_share, "other_kernels": other_kernels,
                         "achieved": round(lane_ops/1e12, 2) if lane_ops else None, "peak": round(VALU_PEAK_LANE_OPS/1e12, 1), "unit": "T lane-instr/s",
                         "frac": round(lane_ops/VALU_PEAK_LANE_OPS, 4) if lane_ops else None,
                         "valu_instructions_per_supersample": round(per_sample, 1) if per_sample else None,
                         "lds_busy": round(lds_busy, 3) if lds_busy else None,
                         "issue_cycles_frac": issue["frac"] if issue else None, "issue_model": issue,
                         "traffic": traffic, "counters_from": counters_from,
                         "launch_ms": round(launch_s*1e3, 3), "frames_per_launch": piece, "launches_per_step": parts,
                         "hbm": {"achieved": round(hbm_achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_achieved/HBM_PEAK_GBS, 4),
                                 "algorithmic_bytes_per_launch": b_alg*piece,
                                 "measured": round(traffic/launch_s/1e9, 1) if traffic else None,
                                 "note": "algorithmic bytes = the reference's two-pass data-flow (SURVEY.md §8d); the fused kernel writes the RGB8 frame only"},
                         "note": "FP32 VALU issue binds the kernel (the 91-tap blur folded to ~300 instructions per supersample, visualizer.frag:36-62's position-only gains once per "
                                 "output pixel for the wave tiles k_visualizer_classify clears — ~400 in all, DESIGN.md §4; LDS ~74 % busy); HBM carries the finished RGB8 frames "
                                 "and L2-resident tables only, so the HBM figure is a fraction of what the two-pass data-flow would move. `frac` = SQ_INSTS_VALU x 64 lanes / launch "
                                 "time / peak counts every instruction ONCE and FALLS whenever the same frame takes fewer instructions (549 in rounds 3-4, 472 in round 5, 403 in "
                                 "round 6: 0.70 -> 0.63 -> 0.58) while frames/s rise; the share of issue SLOTS in use is issue_model.frac_census"},
        }
        if args.scene != "visualizer":
            # the light fragments are bound by the HBM write of the finished frame: the roofline is the FUSED lower bound — W·H·3 bytes
            # per frame, the only bytes a fused kernel has to move (last column) — not the two-pass data-flow
            fused_bytes = float(w*h*3)*piece
            written = fused_bytes/launch_s/1e9
            result["roofline"] = {"bound": "hbm", "kernel": kernel, "kernel_share_of_gpu_time": dominant_share, "other_kernels": other_kernels,
                                  "achieved": round(written, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(written/HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_over_algorithmic": round(traffic/fused_bytes, 3) if traffic else None,
                                  "valu_instructions_per_supersample": round(per_sample, 1) if per_sample else None, "lds_busy": round(lds_busy, 3) if lds_busy else None,
                                  "issue_cycles_frac": issue["frac"] if issue else None, "issue_model": issue, "counters_from": counters_from,
                                  "algorithmic_bytes_per_launch": fused_bytes, "launch_ms": round(launch_s*1e3, 4), "frames_per_launch": piece,
                                  "two_pass_accounting": {"achieved": round(hbm_achieved, 1), "frac": round(hbm_achieved/HBM_PEAK_GBS, 4),
                                                          "note": "the reference's iScreen write + read + iFinal write + read-out; a fused kernel never moves iScreen, so this can exceed 1"},
                                  "note": "launch_ms covers the table kernel + the fused kernel of one step (HIP events on the launch stream)"}
        if baseline is not None:
            result["cpu_baseline"] = baseline
        if export is not None:
            result["export_host"] = export
            result["value_host"] = export["value"]                 # frames/s DELIVERED to a host sink, beside `value` (frames resident in rank 0's HBM)
        if per_rank is not None:
            result["per_rank"] = per_rank
            inbound = sum(r["sent_GB_per_s"] for r in per_rank)
            result["gather"] = {"sdma_dropped": sdma_dropped, "backend": "sdma peer copies (hipIpc windows + SDMA engines named through HSA)" if sdma else dist_backend, "chosen": chosen["transport"],
                                "pieces_per_step": parts, "inbound_GB_per_s_rank0": round(inbound, 2), "loopback": loopback, "loopback_intact": loopback_intact,
                                "legs": [{"transport": leg["transport"], "payload": leg["payload"], "value": leg["value"], "unit": "frames/s",
                                          "ms_per_step": round(leg["elapsed_s"]/max(1, args.steps)*1e3, 3), "loopback_intact": leg["loopback_intact"], "per_rank": leg.get("per_rank")} for leg in legs],
                                "note": "every leg is W untimed + K timed steps of the same frame ranges; `value` is the fastest rgb24 leg (the reference's byte stream resident in "
                                        "rank 0's HBM), the yuv420p legs move half the bytes per link (converted on the rendering rank). sent_GB_per_s = what a rank sent to rank 0 / "
                                        "the timed region: every peer has its own xGMI link to rank 0; a rank whose render_frames_per_s x frame bytes exceeds what its link "
                                        "sustains is link-bound (DESIGN.md §6). peer_copies: the SDMA engines a rank's copies were issued on"}
            planar_legs = [leg for leg in legs if leg["payload"] == "yuv420p" and leg["value"]]
            if planar_legs:
                result["value_yuv420p"] = max(leg["value"] for leg in planar_legs)
    else:
        result = None

    # The JSON line is the LAST line on stdout: libraries that write through C stdio (RCCL prints a version banner that stays in
    # its buffer until exit) are flushed first, on every rank, before rank 0 prints.
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if result is not None:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
