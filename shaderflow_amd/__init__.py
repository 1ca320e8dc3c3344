"""
shaderflow_amd — MI355X-native headless render path with ShaderFlow's Scene/Module/ShaderVariable API.

The per-frame STFT → spectrogram → per-pixel fragment → SSAA resolve → frame read-out path of
BrokenSource/ShaderFlow, rebuilt on hand-written HIP kernels for gfx950 behind a C-ABI
(include/shaderflow_hip.h, shaderflow_amd/libshaderflow_hip.so). This package is the host side: it mirrors the
reference's public classes (same names, fields, lifecycle and error behaviour) so that scenes written for the
reference run unchanged; `install_alias()` additionally publishes it under the name `shaderflow` for scripts
that import the reference's package name (examples/basic/demo.py:6-12).

There is no CPU fallback: importing works anywhere, creating a scene needs the built library and a GPU.
"""
from __future__ import annotations

import importlib
import importlib.machinery
import importlib.util
import logging
import sys
from pathlib import Path

__version__ = "0.1.0"
__about__ = "MI355X-native headless ShaderFlow render path"

logger = logging.getLogger("shaderflow_amd")

package = Path(__file__).parent
"""Path to the package directory"""

resources = package/"resources"
"""Registry stubs that stand where the reference keeps its built-in GLSL (shaderflow/__init__.py:14-15)"""


class _Directories:
    """Where downloaded assets and screenshots go (reference: platformdirs, shaderflow/__init__.py:17-21)"""
    @property
    def user_data_path(self) -> Path:
        path = Path.home()/".local"/"share"/"shaderflow_amd"
        path.mkdir(parents=True, exist_ok=True)
        return path


directories = _Directories()

class _AliasFinder:
    """Resolves `<alias>` and every `<alias>.<sub>` to THE module object of `shaderflow_amd[.<sub>]` — one object under two
    names, so `shaderflow.video.ShaderVideo is shaderflow_amd.video.ShaderVideo` for every submodule, present and future
    (a static list would let a forgotten one be imported a second time as a distinct module)"""

    def __init__(self, alias: str):
        self.alias = alias

    def _real(self, fullname: str):
        if fullname == self.alias or fullname.startswith(self.alias + "."):
            return __name__ + fullname[len(self.alias):]
        return None

    def find_spec(self, fullname, path=None, target=None):
        real = self._real(fullname)
        if real is None:
            return None
        try:
            found = importlib.util.find_spec(real)
        except (ImportError, ValueError):
            found = None
        if found is None:
            return None
        return importlib.machinery.ModuleSpec(fullname, self, is_package=found.submodule_search_locations is not None)

    def create_module(self, spec):
        return importlib.import_module(self._real(spec.name))

    def exec_module(self, module) -> None:
        pass                                                  # already executed under its own name


def install_alias(name: str = "shaderflow") -> None:
    """Make `import shaderflow`, `from shaderflow.scene import ShaderScene`, … resolve to this package"""
    sys.modules[name] = sys.modules[__name__]
    if not any(isinstance(finder, _AliasFinder) and finder.alias == name for finder in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder(name))


def __getattr__(attr: str):
    # Lazy top-level conveniences: shaderflow_amd.ShaderScene etc.
    lazy = {
        "ShaderScene": "scene", "ShaderModule": "module", "ShaderProgram": "shader", "ShaderTexture": "texture",
        "ShaderDynamics": "dynamics", "DynamicNumber": "dynamics", "ShaderMessage": "message",
        "ShaderVariable": "variable", "Uniform": "variable", "ShaderCamera": "camera",
    }
    if attr in lazy:
        import importlib
        return getattr(importlib.import_module(f"{__name__}.{lazy[attr]}"), attr)
    raise AttributeError(attr)
