"""
Shader variables: the declaration objects modules yield from `pipeline()`.
Mirrors shaderflow/variable.py:46-99 of the reference (same names, fields and equality-by-name); with no GLSL
compiler behind it `declaration` is informational, the (type, name, value) triple is what reaches the kernels.
"""
from __future__ import annotations

import copy
from typing import Any, Optional

from attrs import define

GLSL_TYPES = ("sampler2D", "float", "int", "bool", "vec2", "vec3", "vec4", "mat2", "mat3", "mat4")   # variable.py:12-23
DECLARATION_ORDER = ("interpolation", "direction", "qualifier", "type", "name")


@define(eq=False, slots=True)
class ShaderVariable:
    type: str
    name: str
    value: Optional[Any] = None
    qualifier: Optional[str] = None
    direction: Optional[str] = None
    interpolation: Optional[str] = None

    def __hash__(self) -> int:
        return hash(self.name)

    def __eq__(self, other) -> bool:
        return (self.name == other.name)

    def copy(self, **update):
        other = copy.deepcopy(self)
        for key, value in update.items():
            setattr(other, key, value)
        return other

    @property
    def size_string(self) -> Optional[str]:
        return dict(float="f", int="i", bool="i", vec2="2f", vec3="3f", vec4="4f").get(self.type)

    @property
    def declaration(self) -> str:
        parts = (getattr(self, key, None) for key in DECLARATION_ORDER)
        return " ".join(filter(None, parts)).strip() + ";"


@define(eq=False, slots=True)
class Uniform(ShaderVariable):
    qualifier: Optional[str] = "uniform"


@define(eq=False, slots=True)
class InVariable(ShaderVariable):
    direction: Optional[str] = "in"


@define(eq=False, slots=True)
class OutVariable(ShaderVariable):
    direction: Optional[str] = "out"


@define(eq=False, slots=True)
class FlatVariable(ShaderVariable):
    interpolation: Optional[str] = "flat"
