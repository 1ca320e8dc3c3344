"""
ShaderPiano against the reference's per-frame texture contents (tests/golden/piano.npz, written by
tests/golden/make_golden_piano.py from shaderflow/piano/module.py:185-277) and the MIDI reader. CPU only: the module's
textures are replaced by recorders, the DynamicNumbers are the host ones.
"""
import struct
import types
from pathlib import Path

import numpy as np
import pytest

from shaderflow_amd.piano import PianoNote
from shaderflow_amd.piano.midi import read_midi, write_midi
from shaderflow_amd.piano.module import MAX_NOTE, MAX_ROLLING, ShaderPiano

GOLDEN = np.load(Path(__file__).parent/"golden"/"piano.npz")


class Recorder:
    last = None

    def write(self, data=None, **kwargs):
        self.last = np.array(data, copy=True)
        return self

    def clear(self):
        return self


def bare_piano() -> ShaderPiano:
    """A ShaderPiano without a scene: fields at their defaults, textures recording what is written"""
    piano = ShaderPiano.__new__(ShaderPiano)
    for attribute in ShaderPiano.__attrs_attrs__:
        default = attribute.default
        value = default.factory() if hasattr(default, "factory") else default
        object.__setattr__(piano, attribute.name, value)
    piano.scene = types.SimpleNamespace(time=0.0, dt=0.0, realtime=False)
    piano.keys_texture, piano.channel_texture, piano.roll_texture, piano.tempo_texture = Recorder(), Recorder(), Recorder(), Recorder()
    return piano


def test_update_matches_the_reference_frame_by_frame():
    piano = bare_piano()
    for n, start, end, channel, velocity in GOLDEN["notes"]:
        piano.add_note(PianoNote(note=int(n), start=float(start), end=float(end), channel=int(channel), velocity=int(velocity)))
    assert (piano.global_minimum_note, piano.global_maximum_note) == (int(GOLDEN["global_min"]), int(GOLDEN["global_max"]))
    assert piano.duration == float(GOLDEN["duration"])
    fps, frames = float(GOLDEN["fps"]), int(GOLDEN["frames"])
    index, rows = GOLDEN["roll_index"], GOLDEN["roll_rows"]
    time, dt = 0.0, 0.0
    for k in range(frames):
        piano.scene.time, piano.scene.dt = time, dt
        piano.update()
        assert np.array_equal(piano.keys_texture.last.ravel().astype(np.float32), GOLDEN["keys"][k]), k
        assert np.array_equal(piano.channel_texture.last.ravel(), GOLDEN["channels"][k]), k
        assert np.array_equal(np.asarray(piano.note_range_dynamics.value, np.float32), GOLDEN["dynamic"][k]), k
        roll = piano.roll_texture.last
        assert roll.shape == (MAX_NOTE, MAX_ROLLING, 4) and roll.dtype == np.float32
        want = np.zeros_like(roll)
        part = rows[index[k]:index[k + 1]]
        want[part[:, 0].astype(int), part[:, 1].astype(int)] = part[:, 2:]
        assert np.array_equal(roll, want), k
        dt = 1.0/fps
        time += dt
    names = {u.name: u.value for u in piano.pipeline()}
    assert names["iPianoLimit"] == MAX_ROLLING and names["iPianoRollTime"] == GOLDEN["uniforms"][0] and names["iPianoGlobalMin"] == piano.global_minimum_note


def test_notes_between_follows_the_whole_second_buckets():
    piano = bare_piano()
    a = PianoNote(note=60, start=3.1, end=3.2)          # ended earlier in second 3
    b = PianoNote(note=60, start=5.9, end=7.0)
    c = PianoNote(note=60, start=8.5, end=9.0)          # starts after the window
    d = PianoNote(note=61, start=3.5, end=4.0)
    for note in (b, a, c, d):
        piano.add_note(note)
    assert list(piano.notes_between(60, 3.7, 7.7)) == [a, b]           # bucket 3 first (a), then 5 (b); c starts after 7.7
    assert list(piano.notes_between(60, 3.7, 8.6)) == [a, b, c]
    assert list(piano.notes_between(61, 4.2, 4.4)) == [d]               # still in bucket 4 although it ended at 4.0
    assert list(piano.notes_between(62, 0.0, 10.0)) == []
    piano.normalize_velocities(60, 100)
    assert {n.velocity for n in piano.notes} == {80}


def test_midi_round_trip_with_tempo_changes(tmp_path):
    rng = np.random.default_rng(2)
    notes = []
    for group in (0, 1, 2):
        t = 0.0
        for _ in range(30):
            t += float(rng.uniform(0.05, 0.4))
            notes.append(PianoNote(note=int(rng.integers(30, 100)), start=t, end=t + float(rng.uniform(0.05, 0.5)), channel=group,
                                   velocity=int(rng.integers(1, 128))))
    tempo = [(0.0, 120.0), (2.0, 90.0), (5.0, 150.0)]
    path = write_midi(tmp_path/"score.mid", notes, tempo, division=960)
    got, got_tempo = read_midi(path)
    assert len(got) == len(notes)
    assert np.allclose([t for t, _ in got_tempo], [0.0, 2.0, 5.0], atol=1e-3) and np.allclose([b for _, b in got_tempo], [120, 90, 150], rtol=1e-6)
    key = lambda n: (n.channel, round(n.start, 2), n.note)
    for a, b in zip(sorted(got, key=key), sorted(notes, key=key)):
        assert (a.note, a.channel, a.velocity) == (b.note, b.channel, b.velocity)
        assert abs(a.start - b.start) < 2e-3 and abs(a.end - b.end) < 2e-3          # tick quantisation: 960 ppq at >= 90 bpm


def test_midi_running_status_and_note_on_zero(tmp_path):
    # format 0, 96 ppq, default tempo (120 bpm → 1 tick = 1/192 s): running status, note-on velocity 0 as note-off,
    # a program change (new instrument for later notes) and a sysex to skip
    track = bytes([
        0x00, 0x90, 60, 100,            # t=0     on C4
        0x60, 64, 90,                   # t=96    on E4 (running status)
        0x60, 60, 0,                    # t=192   off C4 (velocity 0)
        0x00, 0xF0, 0x03, 1, 2, 0xF7,   # sysex
        0x00, 0xC0, 5,                  # program 5 on channel 0
        0x30, 0x90, 67, 80,             # t=240   on G4
        0x30, 0x80, 64, 0,              # t=288   off E4 — opened under program 0, closed under program 5
        0x30, 67, 0,                    # t=336   off G4 (running status of 0x80)
        0x00, 0xFF, 0x2F, 0x00,
    ])
    raw = b"MThd" + struct.pack(">IHHH", 6, 0, 1, 96) + b"MTrk" + struct.pack(">I", len(track)) + track
    path = tmp_path/"tiny.mid"
    path.write_bytes(raw)
    notes, tempo = read_midi(path)
    assert tempo == [(0.0, 120.0)]
    assert [(n.note, n.velocity) for n in sorted(notes, key=lambda n: n.start)] == [(60, 100), (64, 90), (67, 80)]
    by_pitch = {n.note: n for n in notes}
    assert by_pitch[60].start == 0.0 and by_pitch[60].end == pytest.approx(1.0)
    assert by_pitch[64].start == pytest.approx(0.5) and by_pitch[64].end == pytest.approx(1.5)
    assert by_pitch[67].start == pytest.approx(1.25) and by_pitch[67].end == pytest.approx(1.75)
    assert by_pitch[60].channel == 0 and by_pitch[64].channel == by_pitch[67].channel == 1      # instrument index, not MIDI channel
    with pytest.raises(ValueError):
        (tmp_path/"bad.mid").write_bytes(b"RIFFxxxx")
        read_midi(tmp_path/"bad.mid")
