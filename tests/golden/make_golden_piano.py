#!/usr/bin/env python3
"""
Golden vectors of ShaderPiano: drives the REFERENCE's shaderflow/piano/module.py:185-277 (`update`) with a stand-in
scene (time, dt, realtime) and stand-in textures that record what is written, for a seeded random score, and stores
the score next to the recorded texture contents in piano.npz. Runs only in the build container.

Recorded per frame: the key-press values (keys texture), the channel row, the dynamic note range, and the rolling
texture as its non-zero rows (note, slot, start, end, channel, velocity).
"""
import sys
import types
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden  # noqa: E402,F401  (import shims + the reference on sys.path)

from shaderflow.piano.module import ShaderPiano as RefPiano  # noqa: E402
from shaderflow.piano.notes import PianoNote as RefNote      # noqa: E402


class Recorder:
    def __init__(self):
        self.last = None

    def write(self, data=None, **kwargs):
        self.last = np.array(data, copy=True)
        return self

    def clear(self):
        return self


def score(seed: int, count: int, duration: float):
    rng = np.random.default_rng(seed)
    notes = []
    for _ in range(count):
        start = float(rng.uniform(0, duration))
        length = float(rng.choice([0.02, 0.1, 0.4, 1.3, 2.6])*rng.uniform(0.8, 1.2))
        notes.append((int(rng.integers(36, 96)), start, start + length, int(rng.integers(0, 4)), int(rng.integers(20, 127))))
    # a chord of identical pitches on different channels and two glued notes (release_before_end workaround)
    notes += [(60, 1.0, 2.0, 0, 90), (60, 1.0, 2.5, 1, 70), (60, 1.2, 1.9, 2, 50), (72, 3.0, 3.5, 0, 100), (72, 3.5, 4.0, 0, 110)]
    return np.array(notes, np.float64)


def main():
    fps, seconds = 60.0, 6.0
    notes = score(5, 160, seconds)
    piano = RefPiano.__new__(RefPiano)
    defaults = {a.name: (a.default.factory() if hasattr(a.default, "factory") else a.default) for a in RefPiano.__attrs_attrs__}
    for name, value in defaults.items():
        object.__setattr__(piano, name, value)
    piano.scene = types.SimpleNamespace(time=0.0, dt=0.0, realtime=False)
    piano.keys_texture, piano.channel_texture, piano.roll_texture = Recorder(), Recorder(), Recorder()
    piano.tempo_texture = Recorder()
    for n, start, end, channel, velocity in notes:
        piano.add_note(RefNote(note=int(n), start=float(start), end=float(end), channel=int(channel), velocity=int(velocity)))

    frames = int(fps*seconds)
    keys, chans, dynamic, roll_rows, roll_index = [], [], [], [], [0]
    # the freewheel clock of the scene: dt = 0 on the first frame, then 1/fps (scene.py:456-479 stores dt after the modules ran)
    time, dt = 0.0, 0.0
    for k in range(frames):
        piano.scene.time, piano.scene.dt = time, dt
        piano.update()
        keys.append(piano.keys_texture.last.ravel().copy())
        chans.append(piano.channel_texture.last.ravel().copy())
        dynamic.append(np.array(piano.note_range_dynamics.value, np.float32).copy())
        roll = piano.roll_texture.last
        note_idx, slot_idx = np.nonzero(roll.any(axis=2))
        rows = np.column_stack([note_idx, slot_idx, roll[note_idx, slot_idx]]) if len(note_idx) else np.zeros((0, 6))
        roll_rows.append(rows.astype(np.float32)); roll_index.append(roll_index[-1] + len(rows))
        dt = 1.0/fps
        time += dt
    out = Path(__file__).with_name("piano.npz")
    np.savez_compressed(out, notes=notes, fps=fps, frames=frames,
                        keys=np.array(keys, np.float32), channels=np.array(chans, np.float32), dynamic=np.array(dynamic, np.float32),
                        roll_rows=np.concatenate(roll_rows), roll_index=np.array(roll_index),
                        global_min=piano.global_minimum_note, global_max=piano.global_maximum_note,
                        duration=piano.duration, uniforms=np.array([piano.roll_time, piano.extra_keys, piano.height, piano.black_ratio]))
    print(out.name, out.stat().st_size, "bytes;", frames, "frames;", roll_index[-1], "roll rows")


if __name__ == "__main__":
    main()
