"""Resolution fitting (reference: shaderflow/resolution.py:6-86; its assertions :90-116 are tests/test_host.py)."""
from __future__ import annotations

import builtins
import math
from typing import Optional


class Resolution:

    @classmethod
    def fit(cls,
        old: Optional[tuple] = None,
        new: Optional[tuple] = None,
        max: Optional[tuple] = None,
        ar: Optional[float] = None,
        scale: float = 1.0,
        multiple: int = 2,
    ) -> tuple[int, int]:
        old_width, old_height = (old or (None, None))
        new_width, new_height = (new or (None, None))
        max_width, max_height = (max or (None, None))

        width = (new_width or old_width)
        height = (new_height or old_height)

        if not all((width, height)):
            raise ValueError(f"Can't get a resolution missing component(s): ({width=}, {height=})")

        if (ar is not None):
            from_width = (width, width/ar)
            from_height = (height*ar, height)

            if (new_height is None):
                (width, height) = from_width
            elif (new_width is None):
                (width, height) = from_height
            elif (new_width != old_width):
                (width, height) = from_width
            elif (new_height != old_height):
                (width, height) = from_height
            else:
                (width, height) = from_width

            reduce = builtins.max(
                width/(min(width, max_width or math.inf) or 1),
                height/(min(height, max_height or math.inf) or 1)
            ) or 1
            width, height = (width/reduce, height/reduce)
        else:
            width = min(width, max_width or math.inf)
            height = min(height, max_height or math.inf)

        return (
            multiple*round((width*scale)/multiple),
            multiple*round((height*scale)/multiple),
        )
