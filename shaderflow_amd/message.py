"""
Messages relayed between modules (reference: shaderflow/message.py:6-163). The headless render path only ever
sends the Shader.* messages (shader.py:410-414, texture.py:370-372); the window/mouse/keyboard classes exist so
that scenes which `isinstance`-check them in `handle()` (examples/basic/demo.py:207-211) import unchanged.
"""
from __future__ import annotations

from typing import Any

from attrs import define, field


class ShaderMessage:

    class Custom:
        data: Any = None

    class Mouse:
        @define
        class Position:
            x: int = 0; y: int = 0; dx: int = 0; dy: int = 0
            u: float = 0.0; v: float = 0.0; du: float = 0.0; dv: float = 0.0

        @define
        class Press:
            button: int = 0
            x: int = 0; y: int = 0
            u: float = 0.0; v: float = 0.0

        @define
        class Release(Press):
            pass

        @define
        class Drag(Position):
            pass

        @define
        class Scroll:
            dx: int = 0; dy: int = 0

        @define
        class Enter:
            state: bool = False

    class Window:
        @define
        class Resize:
            width: int = 0; height: int = 0

        @define
        class Iconify:
            state: bool = False

        @define
        class FileDrop:
            files: list = field(factory=list)

            @property
            def first(self):
                return self.files[0] if self.files else None

        class Close:
            pass

    class Keyboard:
        @define
        class Press:
            key: int = 0; action: int = 0; modifiers: int = 0

        @define
        class KeyDown:
            key: int = 0; modifiers: int = 0

        @define
        class KeyUp:
            key: int = 0; modifiers: int = 0

        @define
        class Unicode:
            char: str = ""

    class Shader:
        class RecreateTextures:
            pass

        class Compile:
            pass

        class Render:
            pass
