from .notes import PianoNote


def __getattr__(name):          # ShaderPiano pulls in the texture/device modules: import it on first use
    if name == "ShaderPiano":
        from .module import ShaderPiano
        return ShaderPiano
    raise AttributeError(name)
