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

_SUBMODULES = ("variable", "message", "module", "scheduler", "resolution", "dynamics", "texture", "shader",
               "camera", "exporting", "scene", "tape", "device", "parallel", "synth", "audio", "audio.module", "audio.spectrogram", "audio.waveform",
               "audio.reader", "piano", "piano.notes")


def install_alias(name: str = "shaderflow") -> None:
    """Make `import shaderflow`, `from shaderflow.scene import ShaderScene`, … resolve to this package"""
    import importlib
    sys.modules[name] = sys.modules[__name__]
    for sub in _SUBMODULES:
        sys.modules[f"{name}.{sub}"] = importlib.import_module(f"{__name__}.{sub}")


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
