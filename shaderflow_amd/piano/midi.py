"""
Standard MIDI File reader for ShaderPiano.load_midi.

The reference loads scores through pretty_midi (piano/module.py:167-199), which is not a dependency here; this is a
self-contained SMF (format 0/1) parser that yields what the reference takes from it: notes as
`PianoNote(note, start, end, channel, velocity)` in seconds, where `channel` is the INDEX OF THE INSTRUMENT
(`enumerate(midi.instruments)`: one instrument per (track, MIDI channel, program), numbered in the order their first
note completes, tracks in file order), and the tempo changes as (seconds, bpm).

Rules taken from the SMF specification and pretty_midi's behaviour: delta times are variable-length quantities,
running status applies to channel messages, a note-on with velocity 0 is a note-off, a note-off closes EVERY open
note of its (channel, pitch) that started on an earlier tick, ticks map to seconds through the tempo map of the
whole file (default 120 bpm), SMPTE time division is not supported.
A small writer (`write_midi`) produces files for tests and synthetic scores.
"""
from __future__ import annotations

import struct
from pathlib import Path
from typing import Iterable

from shaderflow_amd.piano.notes import PianoNote


def _vlq(data: bytes, pos: int) -> tuple[int, int]:
    value = 0
    while True:
        byte = data[pos]
        pos += 1
        value = (value << 7) | (byte & 0x7F)
        if not (byte & 0x80):
            return value, pos


def _events(track: bytes):
    """(absolute tick, status, data bytes) of one MTrk chunk; meta events come as status 0xFF with (type, payload)"""
    pos, tick, running = 0, 0, None
    while pos < len(track):
        delta, pos = _vlq(track, pos)
        tick += delta
        status = track[pos]
        if status & 0x80:
            pos += 1
        elif running is None:
            raise ValueError("MIDI track starts with a data byte and no running status")
        else:
            status = running
        if status == 0xFF:
            kind = track[pos]
            length, pos = _vlq(track, pos + 1)
            yield tick, 0xFF, (kind, track[pos:pos + length])
            pos += length
            if kind == 0x2F:
                return
        elif status in (0xF0, 0xF7):
            length, pos = _vlq(track, pos)
            pos += length
        else:
            running = status
            size = 1 if (status & 0xF0) in (0xC0, 0xD0) else 2
            yield tick, status, track[pos:pos + size]
            pos += size


def read_midi(path: Path) -> tuple[list[PianoNote], list[tuple[float, float]]]:
    """(notes, tempo changes (seconds, bpm)) of a Standard MIDI File"""
    raw = Path(path).read_bytes()
    if raw[:4] != b"MThd":
        raise ValueError(f"{path}: not a Standard MIDI File")
    header_length, fmt, ntracks, division = struct.unpack(">IHHH", raw[4:14])
    if division & 0x8000:
        raise ValueError(f"{path}: SMPTE time division is not supported")
    pos, tracks = 8 + header_length, []
    while pos + 8 <= len(raw) and len(tracks) < ntracks:
        tag, size = raw[pos:pos + 4], struct.unpack(">I", raw[pos + 4:pos + 8])[0]
        if tag == b"MTrk":
            tracks.append(list(_events(raw[pos + 8:pos + 8 + size])))
        pos += 8 + size

    # tempo map of the whole file: (tick, microseconds per quarter), the last change on a tick wins
    changes: dict[int, int] = {}
    for events in tracks:
        for tick, status, data in events:
            if status == 0xFF and data[0] == 0x51 and len(data[1]) == 3:
                changes[tick] = int.from_bytes(data[1], "big")
    changes.setdefault(0, 500000)
    ticks = sorted(changes)
    origin, elapsed = [], 0.0                                    # seconds at each tempo change
    for index, tick in enumerate(ticks):
        if index:
            elapsed += (tick - ticks[index - 1])*changes[ticks[index - 1]]/(1e6*division)
        origin.append(elapsed)

    def seconds(tick: int) -> float:
        index = max(k for k, t in enumerate(ticks) if t <= tick)
        return origin[index] + (tick - ticks[index])*changes[ticks[index]]/(1e6*division)

    tempo = [(origin[index], 60e6/changes[tick]) for index, tick in enumerate(ticks)]

    notes: list[PianoNote] = []
    instruments: dict[tuple[int, int, int], int] = {}
    for track_index, events in enumerate(tracks):
        program = [0]*16
        opened: dict[tuple[int, int], list[tuple[int, int]]] = {}
        for tick, status, data in events:
            kind, channel = status & 0xF0, status & 0x0F
            if kind == 0xC0:
                program[channel] = data[0]
            elif kind == 0x90 and data[1] > 0:
                opened.setdefault((channel, data[0]), []).append((tick, data[1]))
            elif kind == 0x80 or (kind == 0x90 and data[1] == 0):
                key = (channel, data[0])
                if key not in opened:
                    continue
                closing = [(t, v) for t, v in opened[key] if t != tick]
                keeping = [(t, v) for t, v in opened[key] if t == tick]
                if closing:
                    instrument = instruments.setdefault((track_index, channel, program[channel]), len(instruments))
                    for started, velocity in closing:
                        notes.append(PianoNote(note=data[0], start=seconds(started), end=seconds(tick), channel=instrument, velocity=velocity))
                if closing and keeping:
                    opened[key] = keeping
                else:
                    del opened[key]
    # pretty_midi lists an instrument's notes together: keep the instruments in order, notes in completion order inside
    notes.sort(key=lambda n: n.channel)
    return notes, tempo


def write_midi(path: Path, notes: Iterable[PianoNote], tempo: Iterable[tuple[float, float]] = ((0.0, 120.0),), division: int = 480) -> Path:
    """Format-1 file: a tempo track, then one track per `channel` value of the notes (MIDI channel = value % 16).
    Times are quantised to ticks under the given tempo map."""
    tempo = sorted(tempo)
    if not tempo or tempo[0][0] != 0.0:
        tempo = [(0.0, 120.0)] + list(tempo)
    marks, tick = [], 0.0                                       # (seconds, tick, bpm)
    for index, (when, bpm) in enumerate(tempo):
        if index:
            tick += (when - tempo[index - 1][0])*tempo[index - 1][1]/60.0*division
        marks.append((when, tick, bpm))

    def to_tick(when: float) -> int:
        base = max((m for m in marks if m[0] <= when), key=lambda m: m[0])
        return int(round(base[1] + (when - base[0])*base[2]/60.0*division))

    def vlq(value: int) -> bytes:
        out = [value & 0x7F]
        while value := value >> 7:
            out.append((value & 0x7F) | 0x80)
        return bytes(reversed(out))

    def chunk(events: list[tuple[int, bytes]]) -> bytes:
        body, last = b"", 0
        for when, payload in sorted(events, key=lambda e: e[0]):
            body += vlq(when - last) + payload
            last = when
        body += vlq(0) + b"\xFF\x2F\x00"
        return b"MTrk" + struct.pack(">I", len(body)) + body

    notes = list(notes)
    groups = sorted({n.channel for n in notes})
    tracks = [chunk([(int(round(m[1])), b"\xFF\x51\x03" + int(round(60e6/m[2])).to_bytes(3, "big")) for m in marks])]
    for group in groups:
        events = []
        for n in (n for n in notes if n.channel == group):
            channel = group % 16
            events.append((to_tick(n.start), bytes([0x90 | channel, n.note, n.velocity])))
            events.append((to_tick(n.end), bytes([0x80 | channel, n.note, 0])))
        # note-offs before note-ons on the same tick, so that glued notes do not close each other
        events.sort(key=lambda e: (e[0], e[1][0] & 0x10))
        tracks.append(chunk(events))
    Path(path).write_bytes(b"MThd" + struct.pack(">IHHH", 6, 1, len(tracks), division) + b"".join(tracks))
    return Path(path)
