#!/usr/bin/env python3
"""
The north star's CPU baseline — "the reference CPU path (numpy FFT + llvmpipe software rasteriser) timed on the host cores with the
core count stated" — measured with THE REFERENCE ITSELF: /root/reference's Visualizer scene (examples/basic/demo.py) exported by
`scene.main()` through Mesa llvmpipe (tests/golden/refhost.py), synthetic sweep + synthetic background as in bench.py, frames piped
to a stand-in encoder that discards them. Build container only (needs /root/reference); the numbers go to BASELINE.md, to
profiles/r03_reference_llvmpipe.txt and, as a static block, into bench.py's JSON line (`cpu_baseline.llvmpipe_container`).

  python tools/measure_reference_cpu.py [--frames-c3 8] [--frames-c2 40]
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT/"tests"/"golden"))


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--frames-c3", type=int, default=8)
    parser.add_argument("--frames-c2", type=int, default=40)
    args = parser.parse_args()
    import refhost
    refhost.install()
    import demo
    from PIL import Image

    from shaderflow_amd import synth
    work = refhost.WORK
    Image.fromarray(synth.background_image(1920, 1080, seed=0)).save(work/"bench_background.png")
    demo.Assets.ethereal = staticmethod(lambda: work/"bench_background.png")
    clip = refhost.write_wav_f32(work/"bench_sweep.wav", synth.sweep_clip(4.0, 44100), 44100)
    context = refhost.Context()
    print(f"# {context.info['GL_VERSION']} | {context.info['GL_RENDERER']} | {os.cpu_count()} logical cores "
          f"(LP_NUM_THREADS={os.environ.get('LP_NUM_THREADS', 'default')})")
    for name, (w, h, ssaa, frames) in {"C2": (1920, 1080, 1, args.frames_c2), "C3": (3840, 2160, 2, args.frames_c3)}.items():
        from shaderflow.audio.spectrogram import BrokenSpectrogram
        BrokenSpectrogram.spectrogram_matrix.cache_clear()
        scene = demo.Visualizer()
        scene.initialize()
        scene.audio._file = clip
        started = time.perf_counter()
        target = work/"measure.rgb"
        scene.main(width=w, height=h, ssaa=ssaa, fps=60.0, time=frames/60.0, output=str(target))
        took = time.perf_counter() - started
        target.unlink()
        print(f"{name}: {w}x{h} ssaa {ssaa}: {frames} frames in {took:.2f} s = {frames/took:.3f} frames/s "
              f"({w*ssaa*h*ssaa*frames/took/1e6:.1f} M supersamples/s; whole scene.main() incl. shader compilation and the encoder pipe)")


if __name__ == "__main__":
    main()
