"""
Soak for the frame tape's two banks and streams: the same clip exported through the tape with batches of 1, 2, 3, 7 and 60 frames
(many bank switches, builds far ahead of renders) must give the frame loop's bytes every time. Run on a GPU box:
    python tools/stress_tape_banks.py [rounds]
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from examples.scenes import MusicBars, Visualizer, Waveform, make   # noqa: E402
from shaderflow_amd import synth                                    # noqa: E402
from shaderflow_amd.tape import FrameTape                           # noqa: E402


def export(cls, pcm, background, batch, frames, w, h, ssaa):
    kwargs = dict(audio=(pcm, 44100))
    if cls is Visualizer:
        kwargs["background"] = background
    scene = make(cls, **kwargs)
    raw = scene.main(width=w, height=h, ssaa=ssaa, fps=60.0, time=frames/60.0, output=bytes, batch=batch)
    return np.frombuffer(raw, np.uint8).reshape(frames, h, w, 3)


def main(rounds: int) -> int:
    frames = 90
    pcm, background = synth.sweep_clip(frames/60.0, 44100), synth.background_image(480, 270, seed=0)
    bad = 0
    for cls, (w, h, ssaa) in ((MusicBars, (320, 180, 2)), (Waveform, (320, 180, 2)), (Visualizer, (384, 216, 2)), (Visualizer, (320, 180, 1))):
        want = export(cls, pcm, background, False, frames, w, h, ssaa)
        for round_ in range(rounds):
            for size in (1, 2, 3, 7, 60):
                FrameTape.BATCH = size
                got = export(cls, pcm, background, True, frames, w, h, ssaa)
                if not np.array_equal(got, want):
                    bad += 1
                    where = np.argwhere((got != want).reshape(frames, -1).any(1)).ravel()
                    print(f"{cls.__name__} {w}x{h} ssaa {ssaa} batch {size} round {round_}: frames {where[:8].tolist()} differ", flush=True)
        print(f"{cls.__name__} {w}x{h} ssaa {ssaa}: {rounds} rounds x 5 batch sizes done", flush=True)
    # large frames: the renders are long, the builds run far ahead of them
    frames = 24
    pcm, background = synth.sweep_clip(frames/60.0, 44100), synth.background_image(1920, 1080, seed=0)
    for cls, (w, h, ssaa) in ((MusicBars, (3840, 2160, 2)), (Visualizer, (3840, 2160, 2)), (Visualizer, (1920, 1080, 1))):
        want = export(cls, pcm, background, False, frames, w, h, ssaa)
        for round_ in range(rounds):
            for size in (2, 5):
                FrameTape.BATCH = size
                got = export(cls, pcm, background, True, frames, w, h, ssaa)
                if not np.array_equal(got, want):
                    bad += 1
                    print(f"{cls.__name__} {w}x{h} ssaa {ssaa} batch {size} round {round_}: differs", flush=True)
        print(f"{cls.__name__} {w}x{h} ssaa {ssaa}: {rounds} rounds x 2 batch sizes done", flush=True)
    FrameTape.BATCH = 60
    print("stress_tape_banks:", "FAILED" if bad else "ok", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 3))
