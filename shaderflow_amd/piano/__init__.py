from .notes import PianoNote
