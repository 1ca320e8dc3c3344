#!/usr/bin/env python3
"""
Golden images of the REFERENCE'S OWN GLSL, executed by an independent OpenGL implementation.

The reference evaluates its fragments inside an OpenGL 3.3 driver (moderngl), which this container does not have —
but it does have Google SwiftShader (OpenGL ES 3.0 on the CPU, shipped inside the `kaleido` wheel). This script
assembles each fragment the way shaderflow/shader.py:190-235 does (declarations, include/shaderflow.glsl,
include/camera.glsl, texture defines, user content; vertex/default.glsl for the vertex stage), reading every GLSL
file from /root/reference at run time, adapts the text MECHANICALLY from `#version 330` to `#version 300 es`
(`to_es` below: GLSL ES has no implicit int→float conversions, so every scalar becomes a float — integer literals
get `.0`, `int`/`ivec` become `float`/`vec`, `int(x)` becomes `trunc(x)`, `%` becomes `mod`, `switch` keeps an int
selector), runs it on the inputs of the parity tests, and stores inputs + rendered RGBA8 images in gles.npz.
No GLSL text is written to the repository; only the images and the numeric inputs are.

What this pins: the oracle's reading of the GLSL semantics (operator precedence, float loop counters, matrix
order, built-ins, the camera chain), the GL sampler (texel addressing, wrap modes, bilinear filtering, unorm8
conversion) and the rasteriser's varyings — against a real GLSL compiler and rasteriser. What it cannot pin to
the bit: `sin/cos/pow/atan` precision and the sub-texel precision of the bilinear filter are implementation
choices (SwiftShader filters with 8 fractional bits), so the comparison is "within a few LSB", stated in the test.
Integer semantics that survive: a division between two names declared `int` becomes `trunc(a / b)` (tetration.frag's
`it / MAX_STEPS`). life/simulation.glsl (int arrays, `%`, ivec2, texelFetch) is compiled WITH its integers and without
the prelude it does not use (`typed_content` below: only float() around the values assigned to colour components).
"""
from __future__ import annotations

import re
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT))

from gles import Context  # noqa: E402
from oracle import binding as O  # noqa: E402
from tests.helpers import visualizer_inputs  # noqa: E402

REF = Path("/root/reference")
SHADERS = REF/"shaderflow/resources/shaders"
EXAMPLES = REF/"examples/basic/shaders"


def _matching_paren(text: str, start: int) -> int:
    depth = 0
    for k in range(start, len(text)):
        depth += (text[k] == "(") - (text[k] == ")")
        if depth == 0:
            return k
    raise ValueError("unbalanced parentheses")


def expand_pasting_macros(text: str) -> str:
    """GLSL ES has no `##`: function-like macros that paste tokens (camera.glsl GetCamera) are expanded and removed"""
    for define in re.finditer(r"^[ \t]*#define[ \t]+(\w+)\((\w+)\)((?:[^\n]*\\\n)*[^\n]*##(?:[^\n]*\\\n)*[^\n]*)\n", text, flags=re.M):
        macro, parameter, body = define.group(1), define.group(2), define.group(3).replace("\\\n", "\n")
        text = text.replace(define.group(0), "")
        def expand(call: re.Match) -> str:
            argument = call.group(1).strip()
            pasted = re.sub(rf"\b{parameter}\s*##\s*(\w+)", lambda m: argument + m.group(1), body)
            return re.sub(rf"\b{parameter}\b", argument, pasted)
        text = re.sub(rf"\b{macro}\(([^()]*)\)", expand, text)
    return text


def to_es(source: str) -> str:
    """GLSL 3.30 text → GLSL ES 3.00 text in which every scalar is a float (see the module docstring)"""
    text = re.sub(r"/\*.*?\*/", "", source, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"^\s*#version[^\n]*\n", "", text, flags=re.M)
    text = expand_pasting_macros(text)
    text = re.sub(r"\bsample\b", "sample_", text)                # reserved in GLSL ES 3.00, an ordinary name in 3.30
    keep: list[str] = []

    def protect(match: re.Match) -> str:
        keep.append(match.group(0))
        return f"@@{len(keep) - 1}@@"

    text = re.sub(r"textureSize\((\w+), 0\)", lambda m: protect(re.match(r".*", f"vec2(textureSize({m.group(1)}, 0))")), text)
    text = re.sub(r"\bcase\s+\d+\s*:", protect, text)
    text = text.replace("gl_InstanceID", "float(gl_InstanceID)")
    # switch (expr) → switch (int(expr))
    out, pos = "", 0
    for match in re.finditer(r"\bswitch\s*\(", text):
        open_at = match.end() - 1
        close_at = _matching_paren(text, open_at)
        out += text[pos:open_at] + "(@@INT@@(" + text[open_at + 1:close_at] + "))"
        pos = close_at + 1
    text = out + text[pos:]
    # integer division between two int variables keeps its meaning once they are floats: a / b → trunc(a / b)
    integers = set(re.findall(r"\bint\s+(\w+)\s*(?:=|;|,)", text))
    text = re.sub(r"\b(\w+)\s*/\s*(\w+)\b", lambda m: f"trunc({m.group(1)} / {m.group(2)})" if {m.group(1), m.group(2)} <= integers else m.group(0), text)
    text = re.sub(r"\bint\s*\(", "trunc(", text)
    text = text.replace("@@INT@@(", "int(")
    text = re.sub(r"\bivec([234])\b", r"vec\1", text)
    text = re.sub(r"\bint\b(?!\()", "float", text)
    text = re.sub(r"(\w+)\s*%\s*(\w+)", r"mod(\1, \2)", text)
    text = re.sub(r"(?<![\w.@])(?<![0-9.][eE][+-])(\d+)(?![\w.@])", r"\1.0", text)
    text = re.sub(r"@@(\d+)@@", lambda m: keep[int(m.group(1))], text)
    # overloads that only differed by int/float now collide: keep the first definition of each signature
    seen, result, pos = set(), "", 0
    pattern = re.compile(r"^[ \t]*(?:\w+)\s+(\w+)\s*\(([^)]*)\)\s*\{", re.M)
    while (match := pattern.search(text, pos)):
        depth, k = 0, match.end() - 1
        while True:
            depth += (text[k] == "{") - (text[k] == "}")
            if depth == 0:
                break
            k += 1
        signature = (match.group(1), tuple(p.strip().rsplit(" ", 1)[0] for p in match.group(2).split(",") if p.strip()))
        result += text[pos:match.start()]
        if signature not in seen:
            seen.add(signature)
            result += text[match.start():k + 1]
        pos = k + 1
    return result + text[pos:]


# the varyings of shader.py:112-124 and the uniforms the pipelines emit (scene.py:687-703, camera.py:196-201,
# audio/module.py:413-421, spectrogram.py:313-320, waveform.py:89-90), all scalars as floats
VARYINGS = ["fragCoord", "stxy", "glxy", "stuv", "astuv", "gluv", "agluv"]
UNIFORMS = {
    "float": ["iTime", "iTau", "iDuration", "iWantAspect", "iQuality", "iSSAA", "iFramerate", "iFrame", "iLayer", "iSubsample",
              "iMouseInside", "iMouse1", "iMouse2", "iCameraMode", "iCameraProjection", "iCameraSeparation", "iCameraZoom",
              "iCameraIsometric", "iCameraFocalLength", "iCameraOrbital", "iCameraDolly", "iAudioVolume", "iAudioVolumeIntegral", "iAudioSTD",
              "iSpectrogramLength", "iSpectrogramBins", "iSpectrogramOffset", "iWaveformLength", "iScreenTemporal", "iScreenLayers"],
    "vec2": ["iResolution", "iMouse", "iScreenSize"],
    "vec3": ["iCameraRight", "iCameraUpward", "iCameraForward", "iCameraPosition", "iCameraZenith"],
    "bool": ["iRealtime"],
}
HEADER = "#version 300 es\nprecision highp float;\nprecision highp int;\nprecision highp sampler2D;\n"


def declarations(stage: str, samplers: list[str]) -> str:
    lines = [f"uniform {kind} {name};" for kind, names in UNIFORMS.items() for name in names]
    lines += [f"uniform sampler2D {name};" for name in samplers]
    if stage == "vertex":
        lines += ["in vec2 vertex_position;", "in vec2 vertex_gluv;", "flat out float instance;"] + [f"out vec2 {v};" for v in VARYINGS]
    else:
        lines += ["out vec4 fragColor;", "flat in float instance;"] + [f"in vec2 {v};" for v in VARYINGS]
    return "\n".join(lines) + "\n"


def history_defines(name: str, temporal: int, layers: int) -> str:
    """texture.py:349-363: plain names for the last layer and the <name>Texture(temporal, layer, astuv) selector"""
    lines = [f"#define {name}{t or ''} {name}{t}x{layers - 1}" for t in range(temporal)]
    lines.append(f"vec4 {name}Texture(int temporal, int layer, vec2 astuv) {{")
    for t in range(temporal):
        for l in range(layers):
            lines += [f"    if (temporal == {t} && layer == {l})", f"        return texture({name}{t}x{l}, astuv);"]
    lines += ["    return vec4(0.0);", "}"]
    return "\n".join(lines) + "\n"


def typed_content(fragment_text: str, uniforms: str) -> str:
    """A fragment that keeps its integer types: comments dropped, scalar colour-component assignments wrapped in float()"""
    text = re.sub(r"/\*.*?\*/", "", fragment_text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"(fragColor\.[rgba]\s*=\s*)([^;]+);", r"\1float(\2);", text)
    varyings = "out vec4 fragColor;\n" + "".join(f"in vec2 {v};\n" for v in VARYINGS)
    return HEADER + uniforms + varyings + text


def build(fragment_text: str, samplers: list[str], extra: str = "") -> tuple[str, str]:
    prelude = (SHADERS/"include/shaderflow.glsl").read_text() + "\n" + (SHADERS/"include/camera.glsl").read_text() + "\n"
    vertex = HEADER + declarations("vertex", samplers) + to_es(prelude + (SHADERS/"vertex/default.glsl").read_text())
    fragment = HEADER + declarations("fragment", samplers) + to_es(prelude + extra + fragment_text)
    return vertex, fragment


def uniform_values(u: O.Uniforms, **more) -> dict:
    values = {}
    for name, _ in u._fields_:
        if name == "user":
            continue
        value = getattr(u, name)
        values[name] = tuple(value) if hasattr(value, "__len__") else value
    values.update(more)
    return values


QUAD = np.array([[-1, -1], [-1, 1], [1, -1], [1, 1]], np.float32)


def main() -> None:
    ctx = Context()
    print(ctx.version)
    out: dict[str, np.ndarray] = {}
    demo = (REF/"examples/basic/demo.py").read_text()
    inline = re.findall(r'\("""(.*?)"""\)', demo, flags=re.S)                # multi_child, multi_main, dynamics, audio

    def run(tag: str, fragment_text: str, u: O.Uniforms, w: int, h: int, textures: dict, params: dict, extra: str = "", **more):
        vertex, fragment = build(fragment_text, list(textures), extra)
        program = ctx.program(vertex, fragment)
        as_bool = lambda p: (p[0] == "linear" if isinstance(p[0], str) else bool(p[0]), bool(p[1]), bool(p[2]))
        handles = {name: ctx.texture(data, *as_bool(params[name])) for name, data in textures.items()}
        image = ctx.draw(program, w, h, uniform_values(u, **more), handles, {"vertex_position": QUAD, "vertex_gluv": QUAD})
        out[f"{tag}.image"] = image
        out[f"{tag}.size"] = np.array([w, h])
        print(f"{tag:28s} {w}x{h} mean {image[..., :3].mean():6.1f}")
        return image

    # --- untextured fragments, several cameras ---------------------------------------------------------------
    cameras = {"plain": {}, "moved": dict(iCameraZoom=1.3, iCameraIsometric=0.2, iCameraPosition=(0.1, -0.05, 0.0)),
               "stereo": dict(iCameraProjection=1, iCameraSeparation=0.07, iCameraZoom=1.2), "equirect": dict(iCameraProjection=2, iCameraZoom=0.8)}
    for cam, kw in cameras.items():
        u = O.default_uniforms(160, 90, iTime=0.75, iTau=0.3, **kw)
        run(f"default.{cam}", (SHADERS/"fragment/default.glsl").read_text(), u, 160, 90, {}, {})
        out[f"default.{cam}.camera"] = np.array([kw.get("iCameraProjection", 0), kw.get("iCameraZoom", 1.0), kw.get("iCameraIsometric", 0.0),
                                                 kw.get("iCameraSeparation", 0.05), *kw.get("iCameraPosition", (0.0, 0.0, 0.0))], np.float64)
    u = O.default_uniforms(96, 54, iTime=3.0, iTau=0.3)
    run("missing", (SHADERS/"fragment/missing.glsl").read_text(), u, 96, 54, {}, {})
    run("shadertoy", (EXAMPLES/"shadertoy.frag").read_text(), u, 96, 54, {}, {})
    run("raymarch", (EXAMPLES/"raymarch.frag").read_text(), O.default_uniforms(160, 90), 160, 90, {}, {})
    run("raymarch.moved", (EXAMPLES/"raymarch.frag").read_text(), O.default_uniforms(160, 90, iCameraPosition=(0.4, 0.2, -1.5), iCameraZoom=0.8), 160, 90, {}, {})
    run("mandelbrot", (REF/"examples/fractals/shaders/mandelbrot.frag").read_text(), O.default_uniforms(160, 90, iQuality=0.2), 160, 90, {}, {})
    run("multi_child", inline[0], O.default_uniforms(64, 36), 64, 36, {}, {})

    # --- audio-reactive fragments on the inputs of the parity tests ---------------------------------------------------
    for volume in (0.0, 0.5, 1.2):
        w, h = 160, 90
        u, arrays, params = visualizer_inputs(w, h, seed=21, volume=volume, bg_size=(120, 68))
        tag = f"visualizer.v{volume}"
        run(tag, (EXAMPLES/"visualizer.frag").read_text(), u, w, h, arrays, params)
        out[f"{tag}.args"] = np.array([21, volume, 120, 68], np.float64)
    w, h = 128, 72
    u, arrays, params = visualizer_inputs(w, h, seed=5)
    arrays["iSpectrogram"] = arrays["iSpectrogram"]*3
    for name in ("bars", "waveform"):
        run(name, (EXAMPLES/f"{name}.frag").read_text(), u, w, h, arrays, params)
    u.user[0] = 0.35
    run("dynamics", inline[2], u, w, h, {"background": arrays["background"]}, params, iShaderDynamics=0.35,
        extra="uniform float iShaderDynamics;\n")

    # --- the sampler alone: one texel grid, every filter / wrap combination, coordinates beyond [0, 1] ---------------------------
    rng = np.random.default_rng(9)
    texels = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    out["sampler.texels"] = texels
    probe = "void main() { fragColor = texture(probe, astuv*2.5 - 0.75); }"
    for linear in (False, True):
        for repeat in (False, True):
            run(f"sampler.{'linear' if linear else 'nearest'}.{'repeat' if repeat else 'clamp'}", probe, O.default_uniforms(70, 50), 70, 50,
                {"probe": texels}, {"probe": (linear, repeat, repeat)})

    # --- layers and history: multipass (both layers), motionblur layer 1, final.glsl ------------------------------------------------
    w, h = 128, 72
    background = rng.integers(0, 256, (54, 96, 3), dtype=np.uint8)
    out["multipass.background"] = background
    u = O.default_uniforms(w, h)
    source = (EXAMPLES/"multipass.frag").read_text()
    defines = history_defines("iScreen", 1, 2)
    layer0 = run("multipass.layer0", source, u, w, h, {"background": background, "iScreen0x0": np.zeros((h, w, 4), np.uint8), "iScreen0x1": np.zeros((h, w, 4), np.uint8)},
                 {"background": (True, True, True), "iScreen0x0": (True, False, False), "iScreen0x1": (True, False, False)}, extra=defines, iLayer=0)
    run("multipass.layer1", source, u, w, h, {"background": background, "iScreen0x0": layer0, "iScreen0x1": np.zeros((h, w, 4), np.uint8)},
        {"background": (True, True, True), "iScreen0x0": (True, False, False), "iScreen0x1": (True, False, False)}, extra=defines, iLayer=1)
    temporal = 4
    history = [rng.integers(0, 256, (54, 96, 4), dtype=np.uint8) for _ in range(temporal)]
    out["motionblur.history"] = np.stack(history)
    textures = {f"iScreen{t}x{l}": history[t] for t in range(temporal) for l in range(2)}
    textures["background"] = background
    run("motionblur.layer1", (EXAMPLES/"motionblur.frag").read_text(), O.default_uniforms(96, 54), 96, 54, textures,
        {name: ((True, True, True) if name == "background" else (True, False, False)) for name in textures},
        extra=history_defines("iScreen", temporal, 2), iLayer=1, iScreenTemporal=temporal, iScreenLayers=2)
    screen = rng.integers(0, 256, (72, 128, 4), dtype=np.uint8)
    out["final.screen"] = screen
    final = (SHADERS/"fragment/final.glsl").read_text().replace("uniform int iSubsample;", "")
    for (fw, fh, sub) in ((64, 36, 2), (64, 36, 1), (128, 72, 2), (32, 18, 4)):
        run(f"final.{fw}x{fh}.k{sub}", final, O.default_uniforms(fw, fh), fw, fh, {"iScreen": screen}, {"iScreen": (True, False, False)}, iSubsample=sub)

    # --- integer semantics: tetration (int division), Conway's life (texelFetch, int arrays, %), its visuals ------------------
    run("tetration", (REF/"examples/fractals/shaders/tetration.frag").read_text(), O.default_uniforms(160, 90), 160, 90, {}, {})
    run("tetration.zoomed", (REF/"examples/fractals/shaders/tetration.frag").read_text(),
        O.default_uniforms(160, 90, iCameraZoom=2.5, iCameraPosition=(-0.7, 0.1, 0.0)), 160, 90, {}, {})
    lw, lh = 48, 27
    states = [rng.integers(0, 2, (lh, lw, 1)).astype(np.float32) for _ in range(5)]
    out["life.states"] = np.stack(states)
    vertex, _ = build("void main() {}", [])
    uniforms = "uniform int iFrame;\nuniform int iLifePeriod;\nuniform vec2 iLifeSize;\nuniform sampler2D iLife1x0;\n"
    simulation = typed_content((EXAMPLES/"life/simulation.glsl").read_text(), uniforms)
    program = ctx.program(vertex, simulation)
    ctx.gl.glUseProgram(program)
    for frame in (0, 6, 7):
        handle = ctx.texture(states[1], False, True, True)
        for name, value in (("iFrame", frame), ("iLifePeriod", 6)):
            ctx.gl.glUniform1i(ctx.gl.glGetUniformLocation(program, name.encode()), value)
        image = ctx.draw(program, lw, lh, {"iLifeSize": (lw, lh), "iResolution": (lw, lh), "iWantAspect": lw/lh}, {"iLife1x0": handle},
                         {"vertex_position": QUAD, "vertex_gluv": QUAD})
        out[f"life_simulation.f{frame}.image"] = image                  # RGBA8 target: the red channel is the cell (0 or 255)
        print(f"life_simulation.f{frame}          alive {int((image[..., 0] > 127).sum())}")
    textures = {f"iLife{t}x0": states[t] for t in range(5)}
    run("life_visuals", (EXAMPLES/"life/visuals.glsl").read_text(), O.default_uniforms(128, 72, iCameraZoom=0.9), 128, 72, textures,
        {name: (False, True, True) for name in textures})

    # --- end to end: the reference's numpy audio state (pipeline.npz, captured from its own code) through its own GLSL:
    #     visualizer.frag at 2x SSAA, then final.glsl — the frames the reference would hand to the encoder ----------------------
    from shaderflow_amd import synth
    P = np.load(HERE/"pipeline.npz")
    fps, frames = float(P["meta"][0]), int(P["meta"][2])
    w, h, ssaa, runtime = 192, 108, 2, frames/fps
    image = synth.background_image(240, 135, seed=7)
    bins = int(P["bins"][0])
    out["frames.index"] = np.array([1, 10, 40, 99])
    out["frames.size"] = np.array([w, h, ssaa])
    vertex, fragment = build((EXAMPLES/"visualizer.frag").read_text(), ["background", "iSpectrogram", "iWaveform"])
    visualizer = ctx.program(vertex, fragment)
    vertex, fragment = build((SHADERS/"fragment/final.glsl").read_text().replace("uniform int iSubsample;", ""), ["iScreen"])
    final = ctx.program(vertex, fragment)
    background = ctx.texture(np.flipud(image), True, True, True)                                  # from_numpy flips (texture.py:327-335)
    for k in out["frames.index"]:
        t = float(P["time"][k])
        u = O.default_uniforms(w, h, iTime=t, iTau=(t/runtime) % 1.0, iDuration=runtime, iSSAA=float(ssaa), iFramerate=fps, iFrame=round(t*fps),
                               iAudioVolume=float(P["vol_value"][k]), iAudioVolumeIntegral=float(P["vol_integral"][k]), iAudioSTD=float(P["std_value"][k]),
                               iSpectrogramLength=1, iSpectrogramBins=bins, iWaveformLength=180)
        spectrogram = ctx.texture(np.ascontiguousarray(P["spec_value"][k]).reshape(bins, 1, 2), False, True, False)   # spectrogram.py:306 texel order
        waveform = ctx.texture(np.ascontiguousarray(P["wave_row"][k]).reshape(1, 180, 2), True, False, False)
        screen = ctx.draw(visualizer, w*ssaa, h*ssaa, uniform_values(u), {"background": background, "iSpectrogram": spectrogram, "iWaveform": waveform},
                          {"vertex_position": QUAD, "vertex_gluv": QUAD})
        frame = ctx.draw(final, w, h, uniform_values(O.default_uniforms(w, h), iSubsample=2), {"iScreen": ctx.texture(screen, True, False, False)},
                         {"vertex_position": QUAD, "vertex_gluv": QUAD})
        out[f"frames.{k}"] = frame[..., :3].copy()
        print(f"frame {k:3d} t={t:.4f} volume {float(P['vol_value'][k]):.4f} mean {frame[..., :3].mean():.1f}")

    np.savez_compressed(HERE/"gles.npz", **out)
    print("gles.npz", (HERE/"gles.npz").stat().st_size, "bytes,", len([k for k in out if k.endswith('.image')]), "images")


if __name__ == "__main__":
    main()
