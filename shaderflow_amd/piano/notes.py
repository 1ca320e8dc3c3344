"""
Equal-temperament note arithmetic.

`ShaderSpectrogram.from_notes` (reference: audio/spectrogram.py:226-245) sizes its filterbank from two notes: one
bin per key between them, edges pushed out by half a semitone. This module provides the `PianoNote` value object the
reference exposes for that (piano/notes.py): MIDI index ↔ frequency ↔ name, with A4 = index 69 = `tuning` Hz.
Only the arithmetic matters to the render path; it is implemented here as a small plain class.
"""
from __future__ import annotations

import math
from typing import Any

PIANO_NOTES = ("C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B")
_BLACK = frozenset((1, 3, 6, 8, 10))
_A4 = 69


def _index_of_frequency(frequency: float, tuning: float) -> int:
    # round-half-even like the reference's built-in round(); 12 semitones per octave around A4
    return round(_A4 + 12.0*math.log2(frequency/tuning))


def _frequency_of_index(index: int, tuning: float) -> float:
    return tuning*2**((index - _A4)/12)


def _index_of_name(name: str) -> int:
    pitch, octave = name[:-1].upper(), int(name[-1])
    return 12*(octave + 1) + PIANO_NOTES.index(pitch)


def _name_of_index(index: int) -> str:
    return f"{PIANO_NOTES[index % 12]}{index//12 - 1}"


class PianoNote:
    """A (possibly timed) MIDI note. `note` is the MIDI index; `frequency` and `name` are views of it."""

    __slots__ = ("note", "start", "end", "channel", "velocity", "tuning")

    def __init__(self, note: int = 60, start: float = 0.0, end: float = 0.0, channel: int = 0,
                 velocity: int = 100, tuning: float = 440):
        self.note, self.start, self.end = note, start, end
        self.channel, self.velocity, self.tuning = channel, velocity, tuning

    def __repr__(self) -> str:
        return f"PianoNote({self.name}, note={self.note}, {self.frequency:.2f} Hz)"

    # constructors ---------------------------------------------------------------------------------------------

    @classmethod
    def from_index(cls, note: int, **kwargs) -> "PianoNote":
        return cls(note=int(note), **kwargs)

    @classmethod
    def from_name(cls, name: str, **kwargs) -> "PianoNote":
        return cls(note=_index_of_name(name), **kwargs)

    @classmethod
    def from_frequency(cls, frequency: float, **kwargs) -> "PianoNote":
        return cls(note=_index_of_frequency(frequency, kwargs.get("tuning", 440)), **kwargs)

    @classmethod
    def get(cls, what: Any, **kwargs) -> "PianoNote":
        """Coerce a PianoNote / MIDI index / name / frequency into a PianoNote"""
        if isinstance(what, PianoNote):
            for key, value in kwargs.items():
                setattr(what, key, value)
            return what
        for kind, build in ((int, cls.from_index), (str, cls.from_name), (float, cls.from_frequency)):
            if isinstance(what, kind):
                return build(what, **kwargs)
        return cls(**kwargs)

    # conversions (static, as the reference exposes them) --------------------------------------------------------

    index_to_name = staticmethod(_name_of_index)
    name_to_index = staticmethod(_index_of_name)

    @staticmethod
    def index_to_frequency(index: int, *, tuning: float = 440) -> float:
        return _frequency_of_index(index, tuning)

    @staticmethod
    def frequency_to_index(frequency: float, *, tuning: float = 440) -> int:
        return _index_of_frequency(frequency, tuning)

    @staticmethod
    def name_to_frequency(name: str, *, tuning: float = 440) -> float:
        return _frequency_of_index(_index_of_name(name), tuning)

    @staticmethod
    def frequency_to_name(frequency: float, *, tuning: float = 440) -> str:
        return _name_of_index(_index_of_frequency(frequency, tuning))

    # views ------------------------------------------------------------------------------------------------------

    @property
    def frequency(self) -> float:
        return _frequency_of_index(self.note, self.tuning)

    @frequency.setter
    def frequency(self, hertz: float) -> None:
        self.note = _index_of_frequency(hertz, self.tuning)

    @property
    def name(self) -> str:
        return _name_of_index(self.note)

    @name.setter
    def name(self, text: str) -> None:
        self.note = _index_of_name(text)

    @staticmethod
    def is_black(note: int) -> bool:
        return (note % 12) in _BLACK

    @staticmethod
    def is_white(note: int) -> bool:
        return (note % 12) not in _BLACK

    @property
    def black(self) -> bool:
        return PianoNote.is_black(self.note)

    @property
    def white(self) -> bool:
        return PianoNote.is_white(self.note)

    @property
    def duration(self) -> float:
        return self.end - self.start

    @duration.setter
    def duration(self, seconds: float) -> None:
        self.end = self.start + seconds
