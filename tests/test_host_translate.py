"""
glsl2hip on the CPU: what the translator rewrites (and what it refuses), the bindings it hands to sfx_program_load, that its
output compiles for gfx950 (hipcc cross-compiles without a GPU), and the GLSL semantics of csrc/jit_runtime.hpp executed on
the host. The rendering of translated fragments is checked on the GPU (tests/test_gpu_translated.py).
"""
import re
import subprocess
from pathlib import Path

import pytest

from shaderflow_amd import glsl2hip as G

ROOT = Path(__file__).resolve().parent.parent
FRAGMENTS = ROOT/"tests"/"golden"/"jit"
CACHE = ROOT/"build"/"jit"


def body(cpp: str) -> str:
    return cpp.split("// ---- translated fragment ----")[1].split("// ---- end of translated fragment ----")[0]


def test_floating_literals_get_a_suffix_and_integers_do_not():
    out = body(G.translate("void main() { float a = 1.0 + .5 + 2. + 1e-4 + 3E5 + 7 + 0x1F + 2u + 1.5f + 4.0lf; }").cpp)
    assert "1.0f + .5f + 2.f + 1e-4f + 3E5f + 7 + 0x1F + 2u + 1.5f + 4.0f" in out
    assert G.translate("void main() { vec2 v1 = gluv; float b = v1.x*2.5; }").cpp.count("v1.x*2.5f") == 1     # `v1.x`: not a literal


def test_parameter_qualifiers_become_values_and_references():
    out = body(G.translate("void f(in float a, out vec3 b, inout float c, const in vec2 d) { b = vec3(a); c += d.x; }\nvoid main() {}").cpp)
    head = re.sub(r"\s+", " ", out.split("{")[0])
    assert "SF_HD void f(float a, vec3& b, float& c, const vec2 d)" in head
    assert "SF_HD void main_()" in out
    arrays = body(G.translate("float total(float values[3], const in vec2 pair[2]) { return values[0] + pair[1].x; }\n"
                              "void fill(out float values[3], inout vec2 p) { values[0] = p.x; }\nvoid main() {}").cpp)
    assert "SF_HD float total(const float (&values)[3], const vec2 (&pair)[2])" in arrays and "SF_HD void fill(float (&values)[3], vec2& p)" in arrays


def test_declarations_of_the_engine_go_and_globals_become_members():
    source = """#version 330
    precision highp float;
    uniform float iLevel;
    uniform vec3 iShade = vec3(1);
    in vec2 stuv;
    layout(location = 0) out vec4 fragColor;
    in vec2 extra_varying;
    float helper(float x);
    float counter = 0;
    const float GAIN = 2.0*1.5;
    const int N = 3;
    const float SCALED = GAIN*N;
    const vec2 CORNER = vec2(1, 0);
    const float ROOT = sqrt(2.0);
    float helper(float x) { return x*GAIN; }
    void main() { fragColor = vec4(helper(iLevel)); }"""
    translation = G.translate(source)
    out = body(translation.cpp)
    assert "#version" not in out and "precision" not in out and "uniform" not in out and "layout" not in out
    assert "vec2 extra_varying;" in out and "in vec2" not in out
    assert out.count("helper") == 2                                             # the prototype is gone: definition and call remain
    assert "static constexpr float GAIN = 2.0f*1.5f;" in out and "static constexpr int N = 3;" in out and "static constexpr float SCALED = GAIN*N;" in out
    assert "const vec2 CORNER = vec2(1, 0);" in out and "static constexpr vec2" not in out
    assert "const float ROOT = sqrt(2.0f);" in out                             # a call: not a constant expression for C++
    assert "float counter = 0;" in out
    names = {(b.name, b.type, b.slot, b.count) for b in translation.bindings}
    assert names == {("iLevel", "float", 0, 1), ("iShade", "vec3", 1, 3)}
    assert "iLevel = user_(0);" in translation.cpp and "iShade = vec3(user_(1), user_(2), user_(3));" in translation.cpp


def test_bindings_follow_the_pipeline_and_skip_what_the_fragment_does_not_read():
    source = """#define background background0x0
    #define iScreen iScreen0x1
    #define iScreen1 iScreen1x1
    void main() { fragColor = texture(background, astuv)*iGain + texture(iSpectrogram, vec2(0, 0)) + float(iCount) + (iFlag ? 1 : 0) + iTime; }"""
    pipeline = [("float", "iTime"), ("sampler2D", "background0x0"), ("sampler2D", "iScreen0x1"), ("sampler2D", "iScreen1x1"), ("sampler2D", "iSpectrogram"),
                ("float", "iGain"), ("int", "iCount"), ("bool", "iFlag"), ("vec4", "iUnused"), ("sampler2D", "iWaveform")]
    translation = G.translate(source, pipeline)
    by_name = {b.name: b for b in translation.bindings}
    assert set(by_name) == {"background0x0", "iSpectrogram", "iGain", "iCount", "iFlag"}      # iScreen*: defined but not reached; iTime: built in
    assert by_name["iSpectrogram"].slot == 1 and by_name["background0x0"].slot == 0 and by_name["background0x0"].sampler
    assert (by_name["iGain"].slot, by_name["iCount"].slot, by_name["iFlag"].slot) == (0, 1, 2)
    assert by_name["iCount"].integer and by_name["iFlag"].integer and not by_name["iGain"].integer
    assert "iCount = user_int_(1);" in translation.cpp and "iFlag = user_int_(2) != 0;" in translation.cpp
    assert translation.cpp.rstrip().endswith("SF_JIT_ENTRY_POINTS(sf::rt::Fragment)") and "#undef background" in translation.cpp
    # the names a scene's pipeline really carries: `<name>0x0` boxes behind `#define <name> <name>0x0` keep the tape's slots too
    real = G.translate("#define iSpectrogram iSpectrogram0x0\n#define iWaveform iWaveform0x0\nvoid main() { fragColor = texture(iSpectrogram, astuv) + texture(iWaveform, astuv); }",
                       [("sampler2D", "iWaveform0x0"), ("sampler2D", "iSpectrogram0x0")])
    assert {b.name: b.slot for b in real.bindings} == {"iSpectrogram0x0": 1, "iWaveform0x0": 2}
    matrix = G.translate("uniform mat3 iMatrix;\nvoid main() { fragColor = vec4(iMatrix[0], 1); }")
    assert [(b.name, b.slot, b.count) for b in matrix.bindings] == [("iMatrix", 0, 9)] and "iMatrix = mat3(user_(0), " in matrix.cpp
    with pytest.raises(G.TranslationError):                                        # 17 x 4 floats: more than the 64-float uniform block
        G.translate("uniform vec4 weights[17];\nvoid main() { fragColor = weights[0]; }")
    with pytest.raises(G.TranslationError):
        G.translate("void main() { fragColor = " + " + ".join(f"texture(s{k}, stuv)" for k in range(17)) + "; }", [("sampler2D", f"s{k}") for k in range(17)])


def test_arrays_casts_keywords_and_discard():
    source = """
    const float WEIGHTS[] = float[](0.25, 0.5, 0.25);
    vec3[2] PAIR = vec3[2](vec3(0), vec3(1));
    struct Ray { vec3 origin; vec3 direction; };
    float new = 1.0;
    void main() {
        float taps[3] = float[3](1.0, 2.0, 3.0);
        int index = int(fragCoord.x) % 3;
        uint bits = uint(index) << 2u;
        Ray ray = Ray(vec3(0), vec3(0, 0, 1));
        bool not = false;
        if (taps[index] > 2.5 || not) discard;
        fragColor = vec4(WEIGHTS[index]*new + float(bits), PAIR[1].xy, ray.direction.z);
    }"""
    out = body(G.translate(source).cpp)
    assert "const float WEIGHTS[3] = {0.25f, 0.5f, 0.25f};" in out
    assert re.search(r"vec3 PAIR\[2\] = \{vec3\(0\), vec3\(1\)\};", out)
    assert "float taps[3] = {1.0f, 2.0f, 3.0f};" in out
    assert "to_int(fragCoord.x) % 3" in out and "to_uint(index) << 2u" in out and "float(bits)" in out
    assert "float new_ = 1.0f;" in out and "bool not_ = false;" in out and "WEIGHTS[index]*new_" in out
    assert "{ discarded_ = true; return; }" in out
    assert "Ray ray = Ray(vec3(0), vec3(0, 0, 1));" in out                     # C++20 initialises the aggregate from parentheses
    assert "{ discarded_ = true; return {}; }" in G.translate("float f() { discard; return 1.0; }\nvoid main() {}").cpp
    with pytest.raises(G.TranslationError):
        G.translate("void main() { fragColor = vec4(0); ")


def test_operators_and_methods_c_plus_plus_does_not_have():
    out = body(G.translate("const float W[3] = float[3](1.0, 2.0, 3.0);\nvoid main() { bool a = stuv.x > 0.5 ^^ stuv.y > 0.5; float s = 0.0;\n"
                           "for (int i = 0; i < W.length(); i++) s += W[i]; fragColor = vec4(a ? s : float(fragColor.length ( ))); }").cpp)
    assert "stuv.x > 0.5f != stuv.y > 0.5f" in out and "i < length_of(W);" in out and "float(length_of(fragColor))" in out
    assert "length(p)" in body(G.translate("void main() { vec2 p = stuv; fragColor = vec4(length(p)); }").cpp)          # the function is left alone
    assert "(a && b) != c" in G.translate("void main() { bool a = true, b = false, c = true; fragColor = vec4(float((a && b) ^^ c)); }").cpp
    with pytest.raises(G.TranslationError):                                        # C++ has no operator between && and ||
        G.translate("void main() { bool a = true, b = false, c = true; fragColor = vec4(float(a && b ^^ c)); }")


UNIFORM_OPTIONS = """
const int TAPS = 3;
uniform float gain = 0.5;
uniform vec3 tint = vec3(1.0, 0.5, 0.25);
uniform vec2 both = vec2(2);
uniform int steps = 3;
uniform bool enabled = true;
uniform float weights[TAPS] = float[](0.25, 0.5, 0.125);
uniform vec2 offsets[2];
void main() {
    float total = 0;
    for (int k = 0; k < TAPS; k++) total += weights[k];
    fragColor = vec4(tint*gain*both.x, total + offsets[1].y + float(steps) + (enabled ? 0.0625 : 0));
}
"""


def test_uniform_initialisers_and_uniform_arrays():
    """GLSL 3.30 §4.3.5: `uniform float gain = 0.5;` reads 0.5 until the host sets it; `uniform T a[N]` is bound element by element"""
    translation = G.translate(UNIFORM_OPTIONS)
    by_name = {b.name: b for b in translation.bindings}
    assert by_name["gain"].default == (0.5,) and by_name["tint"].default == (1.0, 0.5, 0.25) and by_name["both"].default == (2.0, 2.0)
    assert by_name["steps"].default == (3.0,) and by_name["steps"].integer and by_name["enabled"].default == (1.0,)
    assert [by_name[f"weights[{k}]"].default for k in range(3)] == [(0.25,), (0.5,), (0.125,)] and by_name["weights[2]"].array == "weights"
    assert by_name["offsets[1]"].default is None and by_name["offsets[1]"].count == 2 and by_name["offsets[1]"].slot == by_name["offsets[0]"].slot + 2
    assert "float weights[3];" in translation.cpp and "vec2 offsets[2];" in translation.cpp
    for bad in ("uniform float g = sqrt(2.0);\nvoid main() {}", "uniform float w[2] = float[](1.0);\nvoid main() {}",
                "uniform vec3 t = 1.0;\nvoid main() {}", "uniform float m[2][2];\nvoid main() {}"):
        with pytest.raises(G.TranslationError):
            G.translate(bad)


def test_uniform_initialisers_and_arrays_run_on_the_host():
    import numpy as np

    from tests.jit_host import HostFragment
    host = HostFragment(G.translate(UNIFORM_OPTIONS), CACHE)
    pixel = host.render_float(2, 2)[0, 0]
    assert np.allclose(pixel, [1.0, 0.5, 0.25, 0.875 + 0.0 + 3.0 + 0.0625])            # initialisers only; offsets[] starts as zeros
    host.set("gain", 1.0)
    host.set("offsets", [[0.0, 0.0], [0.0, 8.0]])
    host.set("weights", [1.0, 2.0, 3.0])
    host.set("enabled", 0)
    pixel = host.render_float(2, 2)[0, 0]
    assert np.allclose(pixel, [2.0, 1.0, 0.5, 6.0 + 8.0 + 3.0])


def test_fragments_that_take_derivatives_ask_for_the_quad_layout():
    assert "#define SF_JIT_DERIVATIVES 1" in G.translate("void main() { fragColor = vec4(fwidth(stuv.x)); }").cpp
    assert "#define SF_JIT_DERIVATIVES 1" in G.translate("#define slope(v) dFdx(v)\nvoid main() { fragColor = vec4(slope(stuv.x)); }").cpp
    assert "#define SF_JIT_DERIVATIVES 0" in G.translate("void main() { fragColor = vec4(stuv.x); }").cpp


def test_macros_are_rewritten_too_and_removed_afterwards():
    translation = G.translate("#define HALF 0.5\n#define scale(x) ((x)*2.0)\n#if 1\nvoid main() { fragColor = vec4(scale(HALF)); }\n#endif\n")
    assert "#define HALF 0.5f" in translation.cpp and "#define scale(x) ((x)*2.0f)" in translation.cpp
    assert "#undef HALF" in translation.cpp and "#undef scale" in translation.cpp and "#if 1" in translation.cpp


@pytest.mark.parametrize("name", sorted(p.stem for p in FRAGMENTS.glob("*.glsl")))
def test_repository_fragments_translate(name):
    translation = G.translate((FRAGMENTS/f"{name}.glsl").read_text(), [("sampler2D", "background")])
    assert "SF_HD void main_()" in translation.cpp
    assert translation.key == G.translate((FRAGMENTS/f"{name}.glsl").read_text(), [("sampler2D", "background")]).key


def test_translation_compiles_for_gfx950_and_exports_the_entry_points():
    translation = G.translate((FRAGMENTS/"polar.glsl").read_text())
    code = G.compile(translation, cache=CACHE)
    assert code.startswith((b"__CLANG_OFFLOAD_BUNDLE__", b"\x7fELF")) and b"gfx950" in code
    for symbol in (b"sfx_jit_render", b"sfx_jit_fused_1", b"sfx_jit_fused_2", b"sfx_jit_fused_4", b"sfx_jit_layout"):
        assert symbol in code, symbol
    assert (CACHE/f"{translation.key}.hsaco").exists()                         # second call: served from the cache
    assert G.compile(translation, cache=CACHE) == code


def test_compile_errors_carry_the_compiler_message():
    translation = G.translate("void main() { fragColor = undeclared_function(stuv); }")
    with pytest.raises(G.CompileError) as error:
        G.compile(translation, cache=CACHE)
    assert "undeclared_function" in str(error.value)


def test_runtime_header_semantics_on_the_host(tmp_path):
    """tests/jit_runtime_check.hip: swizzles, constructors, implicit conversions, matrix products and the prelude, run on the CPU"""
    binary = tmp_path/"jit_runtime_check"
    build = subprocess.run([G.HIPCC, "--offload-arch=gfx950", "-std=c++20", "-ffp-contract=off", "-O1", f"-I{G.CSRC}",
                            str(ROOT/"tests"/"jit_runtime_check.hip"), "-o", str(binary)], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(binary)], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0 and "all checks passed" in run.stdout, run.stdout + run.stderr


REFERENCE = Path("/root/reference")


@pytest.mark.skipif(not REFERENCE.exists(), reason="the reference checkout is only present in the build container")
def test_every_fragment_of_the_reference_translates():
    """Real-world GLSL in the reference's own style (implicit conversions, float loop counters, macros, switch, texelFetch, structs):
    every fragment it ships goes through the translator (fourteen of them are also compiled, run and compared with the oracle by the
    test_mechanical_translation_* tests below). Read in place, nothing is stored."""
    files = sorted((REFERENCE/"examples").rglob("*.frag")) + sorted((REFERENCE/"examples").rglob("*.glsl")) \
        + sorted((REFERENCE/"shaderflow"/"resources"/"shaders"/"fragment").glob("*.glsl"))
    assert len(files) >= 14
    samplers = ["background", "iSpectrogram", "iWaveform", "child", "iVideo"] + [f"iScreen{t}x{l}" for t in range(4) for l in range(2)] + [f"iLife{t}x0" for t in range(5)]
    pipeline = [("sampler2D", name) for name in samplers] + [("float", "iShaderDynamics"), ("int", "iScreenTemporal"), ("int", "iScreenLayers"),
                                                              ("vec2", "iScreenSize"), ("vec2", "iLifeSize"), ("int", "iLifePeriod"), ("int", "iLifeTemporal")]
    # what ShaderTexture.defines() injects for the history textures (texture.py:349-363)
    defines = [f"#define iScreen{t or ''} iScreen{t}x1" for t in range(4)] + [f"#define iLife{t or ''} iLife{t}x0" for t in range(5)]
    defines.append("vec4 iScreenTexture(int temporal, int layer, vec2 astuv) {")
    for t in range(4):
        for layer in range(2):
            defines += [f"    if (temporal == {t} && layer == {layer})", f"        return texture(iScreen{t}x{layer}, astuv);"]
    defines += ["    return vec4(0.0);", "}"]
    for path in files:
        translation = G.translate("\n".join(defines) + "\n" + path.read_text(), pipeline)
        assert "SF_HD void main_()" in translation.cpp, path.name


# ---- translated fragments executed on the host ---------------------------------------------------------------------------

def test_translated_fragments_on_the_host_match_the_opengl_goldens():
    """The repository's own fragments, translated and run on the CPU, against their OpenGL ES renderings (tests/golden/jit.npz)"""
    import json

    import numpy as np

    from oracle import binding as O
    from tests.jit_host import HostFragment
    golden = np.load(ROOT/"tests"/"golden"/"jit.npz")
    cases = json.loads(str(golden["cases"]))
    for name in ("waves", "cells", "hash", "builtins", "materials", "polar"):
        case = cases[name]
        host = HostFragment(G.translate((FRAGMENTS/f"{name}.glsl").read_text(), [("sampler2D", "background")]), CACHE)
        overrides = {k: (tuple(v) if isinstance(v, list) else v) for k, v in case["uniforms"].items()}
        host.set_uniforms(O.default_uniforms(case["width"], case["height"], **overrides))
        for key, value in {**case["floats"], **case["integers"]}.items():
            host.set(key, value)
        host.bind("background", golden["background"])
        got = host.render(case["width"], case["height"])
        d = np.abs(got.astype(int) - golden[f"{name}.image"].astype(int))
        assert d.max() <= 1 and (d == 0).mean() >= 0.98, (name, int(d.max()), float((d == 0).mean()))


REFERENCE_CASES = {
    # oracle fragment name → (file under /root/reference, textures it samples)
    "default": ("shaderflow/resources/shaders/fragment/default.glsl", ()),
    "missing": ("shaderflow/resources/shaders/fragment/missing.glsl", ()),
    "shadertoy": ("examples/basic/shaders/shadertoy.frag", ()),
    "raymarch": ("examples/basic/shaders/raymarch.frag", ()),
    "tetration": ("examples/fractals/shaders/tetration.frag", ()),
    "mandelbrot": ("examples/fractals/shaders/mandelbrot.frag", ()),
    "bars": ("examples/basic/shaders/bars.frag", ("iSpectrogram",)),
    "waveform": ("examples/basic/shaders/waveform.frag", ("iWaveform",)),
    "visualizer": ("examples/basic/shaders/visualizer.frag", ("background", "iSpectrogram", "iWaveform")),
}


@pytest.mark.skipif(not REFERENCE.exists(), reason="the reference checkout is only present in the build container")
@pytest.mark.parametrize("name", list(REFERENCE_CASES))
def test_mechanical_translation_of_the_reference_equals_the_restatement(name, tmp_path):
    """Two independent derivations of the same pixels: the oracle restates the reference's fragments by hand (and the HIP kernels are
    bit-exact against it on the GPU); here the reference's GLSL text itself goes through the translator and runs on the CPU. Both must
    give the same bytes: they share sfmath's operations, and neither re-associates what the GLSL spells out."""
    import numpy as np

    from oracle import binding as O
    from tests.helpers import oracle_textures, visualizer_inputs
    from tests.jit_host import HostFragment
    path, samplers = REFERENCE_CASES[name]
    w, h = 96, 54
    u, arrays, params = visualizer_inputs(w, h, seed=9)
    u.iTau = 0.3
    translation = G.translate((REFERENCE/path).read_text(), [("sampler2D", s) for s in ("background", "iSpectrogram", "iWaveform")])
    host = HostFragment(translation, tmp_path)
    for sampler in samplers:
        assert host.bind(sampler, arrays[sampler], *params[sampler])
    textures = oracle_textures({k: arrays[k] for k in samplers}, params)
    # the same code object under every camera projection, a moved camera, another time and a louder signal
    settings = [dict(), dict(iCameraProjection=1, iCameraSeparation=0.08, iCameraZoom=1.2), dict(iCameraProjection=2, iCameraZoom=0.7),
                dict(iCameraPosition=(0.3, -0.2, -0.4), iCameraZoom=0.8, iCameraIsometric=0.3, iTime=7.25, iAudioVolume=1.4, iAudioSTD=0.6)]
    for setting in settings:
        for key, value in setting.items():
            if hasattr(value, "__len__"):
                for k, component in enumerate(value):
                    getattr(u, key)[k] = component
            else:
                setattr(u, key, value)
        host.set_uniforms(u)
        got = host.render(w, h)
        want = O.render(name, u, textures, w, h, threads=4)
        d = np.abs(got.astype(int) - want.astype(int))
        # observed: identical bytes, tetration's chaotic boundary included
        assert np.array_equal(got, want), (name, setting, int(d.max()), float((d == 0).mean()))


def _history_defines(name: str, temporal: int, layers: int) -> str:
    """what ShaderTexture.defines() injects (texture.py:349-363): plain names for the last layer and the <name>Texture selector"""
    lines = [f"#define {name}{t or ''} {name}{t}x{layers - 1}" for t in range(temporal)]
    lines.append(f"vec4 {name}Texture(int temporal, int layer, vec2 astuv) {{")
    for t in range(temporal):
        for layer in range(layers):
            lines += [f"    if (temporal == {t} && layer == {layer})", f"        return texture({name}{t}x{layer}, astuv);"]
    return "\n".join(lines + ["    return vec4(0.0);", "}"]) + "\n"


@pytest.mark.skipif(not REFERENCE.exists(), reason="the reference checkout is only present in the build container")
def test_mechanical_translation_of_the_layered_and_temporal_fragments(tmp_path):
    """multipass (both layers), motionblur (history sum through iScreenTexture), life (float state, texelFetch, int arrays, `%`) and
    video: the translated GLSL against the oracle, byte for byte — with the sampler names and helper functions the engine generates"""
    import numpy as np

    from oracle import binding as O
    from tests.jit_host import HostFragment
    shaders = REFERENCE/"examples"/"basic"/"shaders"
    rng = np.random.default_rng(3)
    w, h = 96, 54

    # multipass: layer 0 draws the background, layer 1 reads layer 0 as iScreen0x0 (texture.py:346-347)
    background = rng.integers(0, 256, (40, 64, 3), dtype=np.uint8)
    text = _history_defines("iScreen", 1, 2) + (shaders/"multipass.frag").read_text()
    host = HostFragment(G.translate(text, [("sampler2D", n) for n in ("background", "iScreen0x0", "iScreen0x1")]), tmp_path)
    u = O.default_uniforms(w, h, iLayer=0)
    host.set_uniforms(u)
    assert host.bind("background", background)
    layer0 = host.render(w, h)
    want0 = O.render("multipass", u, {"background": O.make_texture(background)}, w, h, threads=4)
    assert np.array_equal(layer0, want0)
    u.iLayer = 1
    host.set_uniforms(u)
    assert host.bind("iScreen0x0", want0, "linear", False, False)
    want1 = O.render("multipass", u, {"background": O.make_texture(background), 0: O.make_texture(want0, "linear", False, False)}, w, h, threads=4)
    assert np.array_equal(host.render(w, h), want1)

    # motionblur: iScreenTemporal frames of history, layer 1 sums them
    temporal = 4
    history = [rng.integers(0, 256, (h, w, 4), dtype=np.uint8) for _ in range(temporal)]
    text = _history_defines("iScreen", temporal, 2) + (shaders/"motionblur.frag").read_text()
    pipeline = [("sampler2D", f"iScreen{t}x{layer}") for t in range(temporal) for layer in range(2)] + [("int", "iScreenTemporal"), ("sampler2D", "background")]
    host = HostFragment(G.translate(text, pipeline), tmp_path)
    u = O.default_uniforms(w, h, iLayer=1)
    u.user[0] = float(temporal)
    host.set_uniforms(u)
    host.set("iScreenTemporal", temporal)
    for t in range(temporal):
        assert host.bind(f"iScreen{t}x0", history[t], "linear", False, False)
    want = O.render("motionblur", u, {t: O.make_texture(history[t], "linear", False, False) for t in range(temporal)}, w, h, threads=4)
    assert np.array_equal(host.render(w, h), want)

    # life: the simulation renders a float state from the state one frame back; the visuals read five frames
    state = rng.integers(0, 2, (27, 48, 1)).astype(np.float32)
    text = _history_defines("iLife", 5, 1) + (shaders/"life"/"simulation.glsl").read_text()
    pipeline = [("sampler2D", f"iLife{t}x0") for t in range(5)] + [("vec2", "iLifeSize"), ("int", "iLifePeriod")]
    host = HostFragment(G.translate(text, pipeline), tmp_path)
    for frame in (0, 7):
        u = O.default_uniforms(48, 27, iFrame=frame)
        u.user[0], u.user[1], u.user[2] = 48, 27, 6
        host.set_uniforms(u)
        host.set("iLifeSize", (48, 27)); host.set("iLifePeriod", 6)
        assert host.bind("iLife1x0", state, "nearest", True, True)
        want = O.render_to("life_simulation", u, {1: O.make_texture(state, "nearest", True, True)}, 48, 27, 1, np.float32, threads=4)
        assert np.array_equal(host.render_float(48, 27)[..., :1], want), frame
    states = [rng.integers(0, 2, (27, 48, 1)).astype(np.float32) for _ in range(5)]
    text = _history_defines("iLife", 5, 1) + (shaders/"life"/"visuals.glsl").read_text()
    host = HostFragment(G.translate(text, pipeline), tmp_path)
    u = O.default_uniforms(w, h, iCameraZoom=0.9)
    host.set_uniforms(u)
    for t in range(5):
        assert host.bind(f"iLife{t}x0", states[t], "nearest", True, True)
    want = O.render("life_visuals", u, {t: O.make_texture(states[t], "nearest", True, True) for t in range(5)}, w, h, threads=4)
    assert np.array_equal(host.render(w, h), want)

    # video: one texture named after its module (video.py:57-66)
    frame = rng.integers(0, 256, (36, 64, 3), dtype=np.uint8)
    host = HostFragment(G.translate("#define iVideo iVideo0x0\n" + (shaders/"video.frag").read_text(), [("sampler2D", "iVideo0x0")]), tmp_path)
    host.set_uniforms(u)
    assert host.bind("iVideo0x0", frame)
    assert np.array_equal(host.render(w, h), O.render("video", u, {0: O.make_texture(frame)}, w, h, threads=4))


def test_the_translator_asks_for_a_tile_only_where_it_can_pay(monkeypatch):
    """glsl2hip._sampler_worth_a_tile: taps in a loop (or many written out) on one sampler → SF_JIT_TILE_SLOT; a single tap, a fragment
    that takes derivatives, or SHADERFLOW_JIT_TILE=0 → none; SHADERFLOW_JIT_TILE=<name> forces"""
    monkeypatch.delenv("SHADERFLOW_JIT_TILE", raising=False)
    blur = "void main() { vec4 s = vec4(0); for (int k = -3; k <= 3; k++) s += texture(background, astuv + vec2(k, 0)/256.0); fragColor = s/7.0; }"
    single = "void main() { fragColor = texture(background, astuv); }"
    edges = "void main() { vec4 s = vec4(0); for (int k = 0; k < 4; k++) s += texture(background, astuv + k*0.01); fragColor = s*fwidth(astuv.x); }"
    two = ("uniform sampler2D other;\nvoid main() { vec4 s = texture(other, astuv); for (int k = 0; k < 4; k++) s += gtexture(background, gluv*0.5 + k*0.01) "
           "+ texture(other, stuv); fragColor = s; }")
    variables = [("sampler2D", "background")]
    t = G.translate(blur, variables)
    assert t.tiled_sampler == "background" and "#define SF_JIT_TILE_SLOT" in t.cpp
    slot = next(b.slot for b in t.bindings if b.name == "background")
    assert f"#define SF_JIT_TILE_SLOT {slot} " in t.cpp
    assert G.translate(single, variables).tiled_sampler is None
    assert G.translate(edges, variables).tiled_sampler is None
    after = "void main() { float a = 0; for (int k = 0; k < 4; k++) { a += k; } fragColor = a*texture(background, stuv); }"
    assert G.translate(after, variables).tiled_sampler is None           # a loop elsewhere in the text does not make one tap many
    braceless = "void main() { vec4 s = vec4(0); int k = 0; while (k < 4) s += texture(background, astuv + 0.01*k++); fragColor = s; }"
    assert G.translate(braceless, variables).tiled_sampler == "background"
    # helper functions tapping a sampler parameter: the taps count for the sampler a call passes, more when the call is in a loop
    helper = "vec4 blur(sampler2D tex, vec2 uv) { vec4 s = vec4(0); for (int k = -3; k <= 3; k++) s += texture(tex, uv + vec2(k, 0)/64.0); return s/7.0; }\n"
    once = "vec4 one(sampler2D tex, vec2 uv) { return texture(tex, uv); }\n"
    assert G.translate(helper + "void main() { fragColor = blur(background, astuv); }", variables).tiled_sampler == "background"
    assert G.translate(once + "void main() { fragColor = one(background, astuv) + one(background, stuv); }", variables).tiled_sampler is None
    assert G.translate(once + "void main() { vec4 s = vec4(0); for (int k = 0; k < 9; k++) { s += one(background, stuv + 0.01*k); } fragColor = s; }",
                       variables).tiled_sampler == "background"
    assert "SF_JIT_TILE_SLOT" not in G.translate(single, variables).cpp
    assert G.translate(two, variables).tiled_sampler in ("background", "other")
    monkeypatch.setenv("SHADERFLOW_JIT_TILE", "0")
    assert G.translate(blur, variables).tiled_sampler is None
    monkeypatch.setenv("SHADERFLOW_JIT_TILE", "background")
    assert G.translate(single, variables).tiled_sampler == "background"
    assert G.translate(edges, variables).tiled_sampler is None         # derivatives win over the wish


def test_host_build_of_a_tiled_translation_is_the_untiled_fragment(tmp_path, monkeypatch):
    """The tile exists on the device only; the host build of the same unit must compile and shade exactly what the untiled text shades"""
    import numpy as np

    from oracle import binding as O
    from tests.jit_host import HostFragment
    text = "void main() { vec4 s = vec4(0); for (int k = -3; k <= 3; k++) s += texture(background, astuv + vec2(k, k)/64.0); fragColor = s/7.0; }"
    data = np.random.default_rng(3).integers(0, 256, (24, 32, 4), dtype=np.uint8)
    images = []
    for tile in ("1", "0"):
        monkeypatch.setenv("SHADERFLOW_JIT_TILE", tile)
        translation = G.translate(text, [("sampler2D", "background")])
        assert (translation.tiled_sampler is not None) == (tile == "1")
        host = HostFragment(translation, CACHE)
        host.set_uniforms(O.default_uniforms(40, 20))
        host.bind("background", data, "linear", True, False)
        images.append(host.render_float(40, 20))
    assert np.array_equal(images[0].view(np.uint32), images[1].view(np.uint32))
