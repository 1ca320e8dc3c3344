"""
Resolution fitting: "the target is (w, h) — maybe only one of them, maybe with a forced aspect ratio, maybe bounded —
what do we render at?". Same answers as the reference's `Resolution.fit` (shaderflow/resolution.py:9-86), whose
own assertions (:90-116) are part of tests/test_host.py via tests/golden/resolution.npz:

  * a missing new component keeps the old one; nothing known for a component → ValueError;
  * without an aspect ratio each component is clamped to its bound independently;
  * with an aspect ratio the OTHER component follows: from the width when only the width was given or when the
    width is what changed (width wins ties), from the height otherwise; then both shrink by one factor to fit the
    bounds;
  * finally scale and round both to a multiple of `multiple` (2: encoders want even sizes).
"""
from __future__ import annotations

import builtins
import math
from typing import Optional

Pair = Optional[tuple[Optional[float], Optional[float]]]


class Resolution:

    @staticmethod
    def fit(old: Pair = None, new: Pair = None, max: Pair = None, ar: Optional[float] = None,
            scale: float = 1.0, multiple: int = 2) -> tuple[int, int]:
        have_w, have_h = old or (None, None)
        want_w, want_h = new or (None, None)
        cap_w, cap_h = max or (None, None)
        cap_w, cap_h = cap_w or math.inf, cap_h or math.inf

        width, height = want_w or have_w, want_h or have_h
        if not (width and height):
            raise ValueError(f"Can't get a resolution missing component(s): ({width=}, {height=})")

        if ar is None:
            width, height = min(width, cap_w), min(height, cap_h)
        else:
            height_leads = (want_w is None) or (want_h is not None and want_w == have_w and want_h != have_h)
            if height_leads and want_h is not None:
                width = height*ar
            else:
                height = width/ar
            # one common shrink factor keeps the ratio inside the bounds
            shrink = builtins.max(width/min(width, cap_w), height/min(height, cap_h))
            width, height = width/shrink, height/shrink

        def snap(value: float) -> int:
            return multiple*round((value*scale)/multiple)
        return (snap(width), snap(height))
