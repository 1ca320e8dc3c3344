"""
ShaderModule: the base of everything that lives in a scene (reference: shaderflow/module.py:20-178).
Same contract: keyword construction with `scene=`, self-registration in `scene.modules` in creation order,
`build()` at construction, the lifecycle hooks, `relay`, `full_pipeline`, `find`.
"""
from __future__ import annotations

import itertools
import logging
import weakref
from typing import TYPE_CHECKING, Any, Iterable
from weakref import CallableProxyType, ProxyType

from attrs import Factory, define, field

from shaderflow_amd.variable import ShaderVariable

if TYPE_CHECKING:
    from shaderflow_amd.scene import ShaderScene

logger = logging.getLogger("shaderflow_amd")


@define(slots=False, eq=False)
class ShaderModule:

    scene: "ShaderScene" = field(default=None, repr=False)
    """The scene this module belongs to; `ShaderModule(scene=...)` is mandatory for anything but the scene"""

    uuid: int = Factory(itertools.count(1).__next__)
    name: str = None

    def __attrs_post_init__(self):
        from shaderflow_amd.scene import ShaderScene

        # The first module initialised is the scene itself (module.py:38-40)
        if not isinstance(self.scene or self, (CallableProxyType, ProxyType)):
            self.scene = weakref.proxy(self.scene or self)

        if not isinstance(self.scene, ShaderScene):              # module.py:43-47
            raise RuntimeError(
                f"Module of type '{type(self).__name__}' must be added to a 'ShaderScene' instance: "
                f"initialize it with {type(self).__name__}(scene=<ShaderScene>, ...)"
            )

        self.scene.modules.append(self)
        self.commands()

        if not isinstance(self, ShaderScene):
            self.build()

    # lifecycle hooks (module.py:55-116) -----------------------------------------------------------

    def build(self) -> None:
        """Called once, at construction"""

    def setup(self) -> None:
        """Called every time before the main loop"""

    def update(self) -> None:
        """Called every frame"""

    def pipeline(self) -> Iterable[ShaderVariable]:
        return []

    def pipeline_token(self) -> Any:
        """What `pipeline()` would yield, summarised: a value that compares equal as long as every variable this module yields is
        unchanged — a ShaderProgram then skips the module for that frame (shader.py use_scene_pipeline). None (the default, and the
        right answer for any module that cannot tell cheaply): walk `pipeline()` every frame, as the reference does."""
        return None

    def full_pipeline(self) -> Iterable[ShaderVariable]:
        for module in self.scene.modules:
            yield from (module.pipeline() or [])

    def relay(self, message: Any):
        if isinstance(message, type):
            message = message()
        for module in self.scene.modules:
            module.handle(message)
        return self

    def handle(self, message) -> None:
        ...

    def find(self, type: type) -> Iterable["ShaderModule"]:
        for module in self.scene.modules:
            if isinstance(module, type):
                yield module

    @property
    def duration(self) -> float:
        return 0.0

    def ffhook(self, ffmpeg) -> None:
        pass

    def commands(self) -> None:
        ...

    def destroy(self) -> None:
        pass

    def includes(self) -> Iterable[str]:
        yield ""

    def defines(self) -> Iterable[str]:
        yield None

    # logging: `log_info/warn/error/debug/minor(*parts)` prefixed with who is speaking -----------------------------

    @property
    def who(self) -> str:
        return f"(Module {self.uuid:>2} • {type(self).__name__[:12].ljust(12)})"

    def _say(self, level: int, parts: tuple) -> None:
        logger.log(level, " ".join(str(part) for part in (self.who, *parts)))

    def log_info(self, *parts) -> None: self._say(logging.INFO, parts)
    def log_warn(self, *parts) -> None: self._say(logging.WARNING, parts)
    def log_error(self, *parts) -> None: self._say(logging.ERROR, parts)
    def log_debug(self, *parts) -> None: self._say(logging.DEBUG, parts)
    def log_minor(self, *parts) -> None: self._say(logging.DEBUG, parts)
