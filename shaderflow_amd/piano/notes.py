"""PianoNote frequency ↔ note arithmetic (reference: shaderflow/piano/notes.py:9-124), used by
ShaderSpectrogram.from_notes to size the filterbank (spectrogram.py:226-245)."""
from __future__ import annotations

import functools
import math
from typing import Any

from attrs import define

PIANO_NOTES = "C C# D D# E F F# G G# A A# B".split()


@define(eq=False)
class PianoNote:
    note: int = 60
    start: float = 0.0
    end: float = 0.0
    channel: int = 0
    velocity: int = 100
    tuning: float = 440

    @classmethod
    @functools.lru_cache
    def from_index(cls, note: int, **kwargs):
        return cls(note=note, **kwargs)

    @classmethod
    @functools.lru_cache
    def from_name(cls, name: str, **kwargs):
        return cls(note=PianoNote.name_to_index(name), **kwargs)

    @classmethod
    @functools.lru_cache
    def from_frequency(cls, frequency: float, **kwargs):
        return cls(note=PianoNote.frequency_to_index(frequency), **kwargs)

    @classmethod
    def get(cls, object: Any, **kwargs):
        if isinstance(object, PianoNote):
            for key, value in kwargs.items():
                setattr(object, key, value)
            return object
        elif isinstance(object, int):
            return cls.from_index(object, **kwargs)
        elif isinstance(object, str):
            return cls.from_name(object, **kwargs)
        elif isinstance(object, float):
            return cls.from_frequency(object, **kwargs)
        return cls(**kwargs)

    @staticmethod
    def index_to_name(index: int) -> str:
        return f"{PIANO_NOTES[index % 12]}{index//12 - 1}"

    @staticmethod
    def index_to_frequency(index: int, *, tuning: float = 440) -> float:
        return tuning*2**((index - 69)/12)

    @staticmethod
    def name_to_index(name: str) -> int:
        note, octave = name[:-1].upper(), int(name[-1])
        return PIANO_NOTES.index(note) + 12*(octave + 1)

    @staticmethod
    def name_to_frequency(name: str, *, tuning: float = 440) -> float:
        return PianoNote.index_to_frequency(PianoNote.name_to_index(name), tuning=tuning)

    @staticmethod
    def frequency_to_index(frequency: float, *, tuning: float = 440) -> int:
        return round(12*math.log2(frequency/tuning) + 69)

    @staticmethod
    def frequency_to_name(frequency: float, *, tuning: float = 440) -> str:
        return PianoNote.index_to_name(PianoNote.frequency_to_index(frequency, tuning=tuning))

    @property
    def frequency(self) -> float:
        return PianoNote.index_to_frequency(self.note, tuning=self.tuning)

    @frequency.setter
    def frequency(self, value: float):
        self.note = PianoNote.frequency_to_index(value, tuning=self.tuning)

    @property
    def name(self) -> str:
        return PianoNote.index_to_name(self.note)

    @name.setter
    def name(self, value: str):
        self.note = PianoNote.name_to_index(value)

    @staticmethod
    def is_white(note: int) -> bool:
        return (note % 12) in {0, 2, 4, 5, 7, 9, 11}

    @staticmethod
    def is_black(note: int) -> bool:
        return (note % 12) in {1, 3, 6, 8, 10}

    @property
    def white(self) -> bool:
        return PianoNote.is_white(self.note)

    @property
    def black(self) -> bool:
        return PianoNote.is_black(self.note)

    @property
    def duration(self):
        return self.end - self.start

    @duration.setter
    def duration(self, value: float):
        self.end = self.start + value
