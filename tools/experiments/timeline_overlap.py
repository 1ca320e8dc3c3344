#!/usr/bin/env python3
"""
Does the audio tape of batch i + 1 really run BESIDE the render of batch i (MusicBars / Waveform run at 0.62 of
the HBM roof on the bench path against 0.74 for the kernel alone)? Reads a `rocprofv3 --kernel-trace -f csv` trace of a bench.py run and
prints, over the steady part of the run: GPU time per kernel, how much of the render kernels' time another kernel overlaps, and the gaps
in which NO kernel runs (launch latency, stream ordering) — the two ways a step can be longer than its render.

usage: python tools/experiments/timeline_overlap.py TRACE_DIR [render-kernel-prefix ...]
"""
import csv
import sys
from pathlib import Path


def main() -> None:
    root = Path(sys.argv[1])
    render_prefixes = tuple(sys.argv[2:]) or ("k_separable", "k_visualizer_strip", "k_render", "k_resolve", "k_default")
    rows = []
    for table in root.glob("**/*kernel_trace.csv"):
        with open(table) as handle:
            for row in csv.DictReader(handle):
                name = row["Kernel_Name"].replace("sf::", "").replace("void ", "").split("(")[0].strip()
                rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), name, row.get("Queue_Id", "?")))
    rows.sort()
    if not rows:
        raise SystemExit("no kernel trace under " + str(root))
    # the steady part: from the first launch of the last third of the render launches to the end
    renders = [r for r in rows if r[2].startswith(render_prefixes)]
    big = sorted(renders, key=lambda r: r[1] - r[0])[-max(1, len(renders)//3):]
    start = min(r[0] for r in big[len(big)//2:]) if len(big) > 2 else big[0][0]
    window = [r for r in rows if r[0] >= start]
    t0, t1 = window[0][0], max(r[1] for r in window)
    span = t1 - t0
    # union of busy intervals and of render intervals
    def union(intervals):
        total, end = 0, None
        for a, b in sorted(intervals):
            if end is None or a > end:
                total += b - a; end = b
            elif b > end:
                total += b - end; end = b
        return total
    busy = union([(r[0], r[1]) for r in window])
    render_busy = union([(r[0], r[1]) for r in window if r[2].startswith(render_prefixes)])
    other = [(r[0], r[1]) for r in window if not r[2].startswith(render_prefixes)]
    other_busy = union(other)
    both = render_busy + other_busy - busy                           # time in which a render kernel AND another kernel run
    print(f"steady window: {span/1e3:.1f} us, {len(window)} launches on queues {sorted({r[3] for r in window})}")
    print(f"  some kernel runs {busy/span*100:5.1f} % of it;  a render kernel {render_busy/span*100:5.1f} %;  another kernel {other_busy/span*100:5.1f} %;  both at once {both/span*100:5.1f} %;  nothing {100 - busy/span*100:5.1f} %")
    per = {}
    for a, b, name, _ in window:
        entry = per.setdefault(name, [0, 0]); entry[0] += b - a; entry[1] += 1
    for name, (total, count) in sorted(per.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"  {name[:70]:70s} {total/1e3:9.1f} us  {count:5d} launches  {total/count/1e3:8.2f} us each  {total/span*100:5.1f} % of the window")
    import os
    dump = int(os.environ.get("TIMELINE_DUMP", "0"))
    if dump:
        print(f"  first {dump} launches of the window (start, end relative to the window's start in us; queue; kernel):")
        for a, b, name, queue in window[:dump]:
            print(f"    {(a - t0)/1e3:10.1f} {(b - t0)/1e3:10.1f}  q{queue}  {name[:60]}")


if __name__ == "__main__":
    main()
