"""
Shader variables — what a module's `pipeline()` yields and what `ShaderProgram` pushes to a kernel.

API of the reference's shaderflow/variable.py (`ShaderVariable(type, name, value, qualifier, direction,
interpolation)`, `Uniform`, `InVariable`, `OutVariable`, `FlatVariable`, equality and hashing by NAME, `copy(**update)`,
`declaration`, `size_string`). With no GLSL compiler behind it the declaration string is informational; the
(type, name, value) triple is what reaches the device (`sfx_uniform_set` / `sfx_sampler_bind`).
"""
from __future__ import annotations

import copy as _copy
from typing import Any, Optional

GLSL_TYPES = ("sampler2D", "float", "int", "bool", "vec2", "vec3", "vec4", "mat2", "mat3", "mat4")
_VERTEX_FORMAT = {"float": "f", "int": "i", "bool": "i", "vec2": "2f", "vec3": "3f", "vec4": "4f"}
_FIELDS = ("type", "name", "value", "qualifier", "direction", "interpolation")


class ShaderVariable:
    __slots__ = _FIELDS
    _defaults: dict = {}

    def __init__(self, type: str, name: str, value: Optional[Any] = None, qualifier: Optional[str] = None,
                 direction: Optional[str] = None, interpolation: Optional[str] = None):
        # (a scene's pipeline() builds a hundred of these per frame: plain assignments, the class defaults only where nothing was given)
        defaults = self._defaults
        self.type = type if type is not None else defaults.get("type")
        self.name = name if name is not None else defaults.get("name")
        self.value = value if value is not None else defaults.get("value")
        self.qualifier = qualifier if qualifier is not None else defaults.get("qualifier")
        self.direction = direction if direction is not None else defaults.get("direction")
        self.interpolation = interpolation if interpolation is not None else defaults.get("interpolation")

    # Two variables are "the same" when they have the same name: a later pipeline entry replaces an earlier one
    def __eq__(self, other) -> bool:
        return self.name == getattr(other, "name", None)

    def __hash__(self) -> int:
        return hash(self.name)

    def __repr__(self) -> str:
        return f"{type(self).__name__}({self.declaration[:-1]!r}, value={self.value!r})"

    def copy(self, **update) -> "ShaderVariable":
        clone = _copy.deepcopy(self)
        for key, value in update.items():
            setattr(clone, key, value)
        return clone

    @property
    def size_string(self) -> Optional[str]:
        """Vertex-attribute format of the type ("2f" for vec2 …)"""
        return _VERTEX_FORMAT.get(self.type)

    @property
    def declaration(self) -> str:
        """`[interpolation] [in|out] [uniform] type name;`"""
        words = (self.interpolation, self.direction, self.qualifier, self.type, self.name)
        return " ".join(word for word in words if word) + ";"


class Uniform(ShaderVariable):
    __slots__ = ()
    _defaults = {"qualifier": "uniform"}


class InVariable(ShaderVariable):
    __slots__ = ()
    _defaults = {"direction": "in"}


class OutVariable(ShaderVariable):
    __slots__ = ()
    _defaults = {"direction": "out"}


class FlatVariable(ShaderVariable):
    __slots__ = ()
    _defaults = {"interpolation": "flat"}
