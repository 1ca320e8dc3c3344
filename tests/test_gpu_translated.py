"""
Fragments outside the registry on the GPU: GLSL text → glsl2hip.translate → hipcc → sfx_program_load → the generic kernels.

Checked against (1) an independent OpenGL implementation's rendering of the same GLSL (tests/golden/jit.npz, made by
tests/golden/make_golden_jit.py), (2) the restated kernels where a fragment exists in both forms, (3) closed forms through
a whole scene. The fragments are this repository's own (tests/golden/jit/*.glsl).
"""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import binding as O
from shaderflow_amd import _native as N
from shaderflow_amd import glsl2hip
from tests.helpers import Gpu, lsb_report

pytestmark = pytest.mark.gpu
HERE = Path(__file__).parent
G = np.load(HERE/"golden"/"jit.npz")
CASES = json.loads(str(G["cases"]))
# the same fragments as `scene.shader.fragment` of a scene of the reference itself, rendered on desktop OpenGL (Mesa llvmpipe:
# tests/golden/make_golden_jit_mesa.py; even sizes only — scene.main() fits resolutions to even numbers)
M = np.load(HERE/"golden"/"jit_mesa.npz")
MESA_CASES = json.loads(str(M["cases"]))
CACHE = HERE.parent/"build"/"jit"


@pytest.fixture()
def gpu():
    g = Gpu()
    yield g
    g.close()


def load(gpu: Gpu, text: str, variables=()) -> tuple[N.Handle, glsl2hip.Translation]:
    translation = glsl2hip.translate(text, variables)
    code = glsl2hip.compile(translation, cache=CACHE)
    names = [b.name.encode() for b in translation.bindings]
    table = (N.Binding*max(1, len(names)))(*[N.Binding(n, int(b.sampler), b.slot, b.count, int(b.integer)) for n, b in zip(names, translation.bindings)])
    handle = N.Handle()
    N.check(gpu.lib.sfx_program_load(gpu.ctx.handle, code, len(code), table, len(names), C.byref(handle)))
    return handle, translation


def render_case(gpu: Gpu, name: str) -> np.ndarray:
    case = CASES[name]
    text = (HERE/"golden"/"jit"/f"{name.split('.')[0]}.glsl").read_text()
    prog, _ = load(gpu, text, [("sampler2D", "background")])
    overrides = {k: (tuple(v) if isinstance(v, list) else v) for k, v in case["uniforms"].items()}
    gpu.set_uniforms(prog, O.default_uniforms(case["width"], case["height"], **overrides))
    for key, value in case["floats"].items():
        assert gpu.set_values(prog, key, value), key
    for key, value in case["integers"].items():
        assert gpu.set_values(prog, key, value, integer=True), key
    gpu.bind(prog, "background", gpu.texture(G["background"], "linear", True, True))
    image = gpu.render(prog, case["width"], case["height"])
    N.check(gpu.lib.sfx_program_destroy(prog))
    return image


@pytest.mark.parametrize("name", list(CASES))
def test_translated_fragment_against_an_opengl_implementation(gpu, name):
    got = render_case(gpu, name)
    want = G[f"{name}.image"]
    d = np.abs(got.astype(int) - want.astype(int))
    print(name, lsb_report(got, want))
    # differences of neighbouring pixels scaled by the resolution amplify the last bit of the built-ins: more 1 LSB values there
    assert d.max() <= 1 and (d == 0).mean() >= (0.95 if name.startswith("edges") else 0.98), lsb_report(got, want)


@pytest.mark.parametrize("name", list(MESA_CASES))
def test_translated_fragment_against_the_reference_on_desktop_opengl(gpu, name):
    """The translated code object against what the REFERENCE renders with the same text as its fragment (its own source assembly,
    `#version 330`, Mesa llvmpipe): within 1 LSB on every value"""
    got = render_case(gpu, name)
    want = M[f"{name}.image"]
    d = np.abs(got.astype(int) - want.astype(int))
    print(name, lsb_report(got, want))
    assert d.max() <= 1 and (d == 0).mean() >= 0.90, lsb_report(got, want)


GRADIENT = """
void main() {
    fragColor = vec4(stuv.x, 1 - stuv.x, iTime/2, 1);
}
"""


def test_fused_resolve_of_a_translated_fragment_equals_render_then_resolve(gpu):
    """sfx_render_resolve on a loaded program (its own code object's fused kernels) against sfx_render + sfx_resolve"""
    text = (HERE/"golden"/"jit"/"waves.glsl").read_text()
    prog, _ = load(gpu, text, [("sampler2D", "background")])
    gpu.bind(prog, "background", gpu.texture(G["background"], "linear", True, True))
    for (ssaa, subsample) in ((1, 1), (2, 2), (2, 1), (4, 4), (4, 2)):
        w, h = 100, 58
        u = O.default_uniforms(w, h, iTime=0.4, iSSAA=float(ssaa))
        gpu.set_uniforms(prog, u)
        fused = gpu.render_resolve(prog, w, h, ssaa, subsample)
        screen = gpu.render(prog, w*ssaa, h*ssaa)
        two_pass = gpu.resolve(screen, w, h, subsample)
        d = np.abs(fused.astype(int) - two_pass.astype(int))
        assert d.max() <= 1, ((ssaa, subsample), lsb_report(fused, two_pass))
    N.check(gpu.lib.sfx_program_destroy(prog))


def test_implicit_conversions_compute_what_the_explicit_text_computes(gpu):
    """GLSL 3.30 converts int to float implicitly (§4.1.10); the translation must give the same bits as the text that spells the floats out"""
    loose = """
    const int N = 3;
    float falloff(float x, int power) { float r = 1; for (int i = 0; i < power; i++) r *= x; return r; }
    void main() {
        vec3 c = vec3(1, 11, 26)/255;
        vec2 p = gluv*2 - 1/2;
        c += max(p.x, 0) + clamp(p.y, 0, 1) + pow(abs(p.x), 2) + mix(0, 1, stuv.y) + smoothstep(0, 2, length(p)) + step(1, p.x);
        c *= 1 - 0.25*falloff(stuv.x, N);
        c.rg += mod(p, 2)/4;
        float k = 3;
        c.b += k/4 + float(N)/8 + N/2;
        fragColor = vec4(c/(1 + c), 1);
    }"""
    strict = """
    const int N = 3;
    float falloff(float x, int power) { float r = 1.0; for (int i = 0; i < power; i++) r *= x; return r; }
    void main() {
        vec3 c = vec3(1.0, 11.0, 26.0)/255.0;
        vec2 p = gluv*2.0 - float(1/2);
        c += max(p.x, 0.0) + clamp(p.y, 0.0, 1.0) + pow(abs(p.x), 2.0) + mix(0.0, 1.0, stuv.y) + smoothstep(0.0, 2.0, length(p)) + step(1.0, p.x);
        c *= 1.0 - 0.25*falloff(stuv.x, N);
        c.rg += mod(p, 2.0)/4.0;
        float k = 3.0;
        c.b += k/4.0 + float(N)/8.0 + float(N/2);
        fragColor = vec4(c/(1.0 + c), 1.0);
    }"""
    images = []
    for text in (loose, strict):
        prog, _ = load(gpu, text)
        gpu.set_uniforms(prog, O.default_uniforms(96, 54))
        images.append(gpu.render(prog, 96, 54, comps=4, dtype=np.float32))
        N.check(gpu.lib.sfx_program_destroy(prog))
    assert np.array_equal(images[0], images[1]) and images[0].std() > 0.01


def test_translated_gradient_is_the_closed_form(gpu):
    prog, _ = load(gpu, GRADIENT)
    w, h = 64, 8
    gpu.set_uniforms(prog, O.default_uniforms(w, h, iTime=1.0))
    got = gpu.render(prog, w, h, comps=4, dtype=np.float32)
    x = ((np.arange(w, dtype=np.float32) + np.float32(0.5))/np.float32(w)).astype(np.float32)
    aspect = np.float32(w)/np.float32(h)
    stuv_x = ((((x*np.float32(2) - np.float32(1))*aspect) + np.float32(1))/np.float32(2)).astype(np.float32)
    assert np.array_equal(got[3, :, 0], stuv_x) and np.array_equal(got[3, :, 1], np.float32(1) - stuv_x)
    assert (got[..., 2] == 0.5).all() and (got[..., 3] == 1).all()
    N.check(gpu.lib.sfx_program_destroy(prog))


def test_a_code_object_built_against_other_headers_is_refused(gpu):
    translation = glsl2hip.translate(GRADIENT)
    code = bytearray(glsl2hip.compile(translation, cache=CACHE))
    handle = N.Handle()
    with pytest.raises(N.NativeError):
        N.check(gpu.lib.sfx_program_load(gpu.ctx.handle, bytes(code[:4096]), 4096, None, 0, C.byref(handle)))      # truncated: not a code object


def test_a_code_object_with_another_argument_layout_is_refused(gpu, monkeypatch):
    """Same size or not: a code object compiled against a RenderArgs whose members sit elsewhere (here: the profiling build's extra
    member) carries another layout fingerprint and must not be launched with this library's arguments"""
    translation = glsl2hip.translate(GRADIENT)
    monkeypatch.setattr(glsl2hip, "FLAGS", [*glsl2hip.FLAGS, "-DSF_SECTION_TIMERS"])
    monkeypatch.setattr(glsl2hip, "_fingerprint", None)
    code = glsl2hip.compile(translation, cache=CACHE)
    monkeypatch.setattr(glsl2hip, "_fingerprint", None)
    handle = N.Handle()
    assert gpu.lib.sfx_program_load(gpu.ctx.handle, code, len(code), None, 0, C.byref(handle)) != N.OK
    assert b"argument layout" in gpu.lib.sfx_last_error()


from tests.helpers import SCROLL_FRAGMENT as SCROLL  # noqa: E402


@pytest.mark.parametrize("smooth", [False, True])
def test_scrolling_spectrogram_frame_loop_tape_and_replay(gpu, smooth):
    """ShaderSpectrogram with length > 0 (spectrogram.py:272-274, 298-311): a `length*fps` columns wide texture, one column rewritten
    per frame at (offset+1) % width, iSpectrogramOffset telling the fragment where. (1) the frame tape — one texture state per frame
    of a batch, k_spectrogram_scroll — gives the frame loop's bytes across batch boundaries and ring wrap-arounds; (2) the frame loop
    is what a replay of the reference's update rule on the oracle's audio tape, rendered by the HOST build of the same translation,
    produces."""
    from shaderflow_amd import ShaderScene, synth
    from shaderflow_amd.audio import ShaderAudio
    from shaderflow_amd.audio.spectrogram import ShaderSpectrogram
    from shaderflow_amd.piano import PianoNote
    from shaderflow_amd.tape import FrameTape
    from tests.jit_host import HostFragment

    pcm, sr = synth.sweep_clip(2.0, 44100), 44100
    w, h, fps, frames, length = 96, 54, 60.0, 100, 0.5
    width = int(length*fps)

    class Scroller(ShaderScene):
        def build(self):
            super().build()
            self.audio = ShaderAudio(scene=self, name="iAudio")
            self.audio.load(samples=pcm, samplerate=sr)
            self.spectrogram = ShaderSpectrogram(scene=self, audio=self.audio, length=length, smooth=smooth)
            self.spectrogram.from_notes(start=PianoNote.from_frequency(20), end=PianoNote.from_frequency(14000), piano=True)
            self.shader.fragment = SCROLL

    probe = Scroller()
    probe.initialize()
    assert FrameTape.applicable(probe) and probe.spectrogram.length_samples == width
    for ssaa in (1, 2):
        kw = dict(width=w, height=h, fps=fps, time=frames/fps, ssaa=ssaa, output=bytes)
        loop = np.frombuffer(Scroller().main(batch=False, **kw), np.uint8).reshape(frames, h, w, 3)
        tape = np.frombuffer(Scroller().main(batch=None, **kw), np.uint8).reshape(frames, h, w, 3)
        assert np.array_equal(loop, tape), lsb_report(tape, loop)
        assert loop[-1].std() > 5

    # replay of spectrogram.py:298-311 on the oracle's tape (ssaa 1 → final.glsl's 3x3 tent over the shaded frame)
    planar = np.ascontiguousarray(pcm.T)
    times, dts, rdts = O.clock(fps, frames)
    _, tell = O.reader(rdts, sr, 2, planar.shape[1])
    fmin, fmax, bins = O.from_notes(O.lib().sfo_note_of_frequency(20.0, 440.0), O.lib().sfo_note_of_frequency(14000.0, 440.0), True)
    indptr, indices, data = O.filterbank(0, 0, fmin, fmax, bins, 12, sr)
    spec = O.DynF32(2*bins, 4, 1, 0)
    texture = np.zeros((bins, width, 2), np.float32)
    host = HostFragment(glsl2hip.translate(SCROLL, [("sampler2D", "iSpectrogram")]), CACHE)
    offset = 0
    kw = dict(width=w, height=h, fps=fps, time=frames/fps, ssaa=1, output=bytes)
    loop = np.frombuffer(Scroller().main(batch=False, **kw), np.uint8).reshape(frames, h, w, 3)
    check = {0, 1, width - 2, width - 1, width, 59, 60, 61, frames - 1}
    for k in range(frames):
        offset = (offset + 1) % width
        target = O.csr_dot(indptr, indices, data, O.fft_power(planar, int(tell[k])))
        texture[:, offset, :] = spec.step(target.ravel(), abs(dts[k])).reshape(bins, 2)
        if k not in check:
            continue
        u = O.default_uniforms(w, h, iTime=times[k], iTau=(times[k]/(frames/fps)) % 1.0, iDuration=frames/fps, iDeltatime=dts[k], iFramerate=fps,
                               iFrame=round(times[k]*fps), iSubsample=2, iSpectrogramLength=width, iSpectrogramBins=bins, iSpectrogramSmooth=int(smooth),
                               iSpectrogramOffset=offset/width)
        host.set_uniforms(u)
        host.bind("iSpectrogram", texture.copy(), "linear" if smooth else "nearest", True, False)
        want = O.resolve(host.render(w, h), w, h, 2)
        d = np.abs(loop[k].astype(int) - want.astype(int))
        assert d.max() <= 1, (k, lsb_report(loop[k], want))


def test_scene_with_its_own_fragment_through_the_frame_tape(gpu):
    """A stock scene with a fragment of its own batches through the clock tape (iTime/iFrame per frame on the device) like a registry
    fragment does: same bytes as the frame loop, fused and two-pass"""
    from shaderflow_amd import ShaderScene
    from shaderflow_amd.tape import FrameTape

    class Pulse(ShaderScene):
        def build(self):
            super().build()
            self.shader.fragment = "void main() { vec2 p = gluv*rotate2d(iTime); fragColor = vec4(0.5 + 0.5*sin(4.0*p.x + iTime), fract(float(iFrame)/16.0), stuv.y, 1.0); }"

    probe = Pulse()
    probe.initialize()
    assert FrameTape.applicable(probe)
    for ssaa in (1, 2):
        kw = dict(width=96, height=54, fps=60, time=70/60, ssaa=ssaa, output=bytes)
        loop = np.frombuffer(Pulse().main(batch=False, **kw), np.uint8).reshape(70, 54, 96, 3)
        scene = Pulse()
        tape = np.frombuffer(scene.main(batch=None, **kw), np.uint8).reshape(70, 54, 96, 3)
        assert scene.shader.kernel == "translated"
        assert np.array_equal(loop, tape), lsb_report(tape, loop)
        assert [int(f[0, 0, 1]) for f in tape[:4]] == [int(np.rint(np.float32(k)/np.float32(16)*255)) for k in range(4)]


def test_scene_with_its_own_fragment_and_uniforms(gpu):
    """A scene whose fragment and uniforms exist nowhere in the registry (frame loop: its pipeline is python): the frames are the
    closed form of the fragment"""
    from shaderflow_amd import ShaderScene
    from shaderflow_amd.variable import Uniform

    class Custom(ShaderScene):
        def build(self):
            super().build()
            self.shader.fragment = "void main() { fragColor = vec4(iLevel*stuv.x, iShade.g, float(iFrame)/8.0, 1.0); }"

        def pipeline(self):
            yield from ShaderScene.pipeline(self)
            yield Uniform("float", "iLevel", 0.5)
            yield Uniform("vec3", "iShade", (0.1, 0.6, 0.3))

    scene = Custom()
    raw = scene.main(width=64, height=36, fps=30, time=4/30, output=bytes)
    frames = np.frombuffer(raw, np.uint8).reshape(-1, 36, 64, 3)
    assert scene.shader.kernel == "translated" and len(frames) == 4
    assert (frames[..., 1] == 153).all()                                       # 0.6*255
    assert [int(f[0, 0, 2]) for f in frames] == [int(np.rint(np.float32(k)/np.float32(8)*255)) for k in range(4)]
    assert frames[0, 0, -1, 0] > frames[0, 0, 0, 0]


def test_scenes_with_several_translated_programs_layers_and_history(gpu):
    """What the reference's MultiShader and Multipass/MotionBlur scenes do, with fragments of this repository: a second program whose
    texture is sampled by name, a layered target, and a temporal history read back — frames follow the closed forms"""
    from shaderflow_amd import ShaderProgram, ShaderScene

    class TwoPrograms(ShaderScene):
        def build(self):
            super().build()
            self.child = ShaderProgram(scene=self, name="child")
            self.child.fragment = "void main() { fragColor = vec4(astuv.x, 0.25, 0.5, 1.0); }"
            self.shader.fragment = "void main() { fragColor = vec4(texture(child, astuv).rgb*vec3(1.0, 2.0, 0.5), 1.0); if (astuv.y > 0.75) discard; }"

    frames = np.frombuffer(TwoPrograms().main(width=64, height=32, fps=30, time=2/30, output=bytes), np.uint8).reshape(2, 32, 64, 3)
    # rows of `output=bytes` are GL's, bottom-up: the top quarter was discarded (final.glsl at ssaa 1, subsample 2 blends the rows next to the edge)
    kept = frames[0][:23]
    assert (frames[0][25:] == 0).all()
    assert (kept[..., 1] == 128).all() and (kept[..., 2] == 64).all() and kept[0, -1, 0] > 250 and kept[0, 0, 0] < 4

    class Feedback(ShaderScene):
        """layer 0 draws a bar that moves with the frame number, layer 1 averages it with what layer 1 showed one frame ago"""
        def build(self):
            super().build()
            self.shader.texture.layers = 2
            self.shader.texture.temporal = 2
            self.shader.fragment = (
                "void main() {\n"
                "    if (iLayer == 0) { fragColor = vec4(vec3(step(float(iFrame)/4.0, astuv.x)), 1.0); return; }\n"
                "    vec4 now = texture(iScreen0x0, astuv);\n"
                "    vec4 before = iScreenTexture(1, 1, astuv);\n"
                "    fragColor = vec4(0.5*now.rgb + 0.5*before.rgb, 1.0);\n"
                "}\n")

    scene = Feedback()
    frames = np.frombuffer(scene.main(width=64, height=32, fps=30, time=6/30, subsample=1, output=bytes), np.uint8).reshape(6, 32, 64, 3)   # subsample 1: final.glsl is a copy
    assert scene.shader.kernel == "translated"
    # iFinal shows the frame rendered temporal-1 = 1 frame earlier (DESIGN.md "Layered and temporal targets"); the first is black
    assert (frames[0] == 0).all()
    row = frames[:, 16, :, 0].astype(int)
    expected_previous = np.zeros(64)
    for k in range(5):                                                          # frame k of the render loop is shown as output k+1
        bar = ((np.arange(64) + 0.5)/64 >= k/4.0).astype(float)
        value = 0.5*bar + 0.5*expected_previous
        assert np.abs(row[k + 1] - np.rint(value*255)).max() <= 1, k
        expected_previous = np.rint(value*255)/255


def test_array_parameters_matrix_uniforms_and_discard_in_a_function(gpu):
    text = """
    uniform mat2 iTurn;
    uniform mat3 iMix;
    float total(float values[3]) { return values[0] + values[1] + values[2]; }
    void fill(out float values[3], float seed) { for (int i = 0; i < 3; i++) values[i] = seed*float(i + 1); }
    float keep(vec2 p) { if (p.x > 0.5) discard; return 0.25; }
    void main() {
        float values[3];
        fill(values, 0.1);
        vec2 p = iTurn*vec2(1.0, 0.0);
        vec3 c = iMix*vec3(1.0, 2.0, 3.0);
        fragColor = vec4(total(values), p.y, c.z/32.0, keep(astuv));
    }"""
    prog, translation = load(gpu, text)
    gpu.set_uniforms(prog, O.default_uniforms(32, 8))
    turn = np.array([0.0, 1.0, -1.0, 0.0], np.float32)                            # columns (0, 1) and (-1, 0): a quarter turn
    mix = np.arange(1, 10, dtype=np.float32)                                      # columns (1,2,3) (4,5,6) (7,8,9)
    for name, code, values in (("iTurn", N.T_MAT2, turn), ("iMix", N.T_MAT3, mix)):
        known = C.c_int()
        N.check(gpu.lib.sfx_uniform_set(prog, name.encode(), code, values.ctypes.data, C.byref(known)))
        assert known.value
    got = gpu.render(prog, 32, 8, comps=4, dtype=np.float32)
    left, right = got[:, :16], got[:, 16:]
    assert np.allclose(left[..., 0], 0.6, atol=1e-6) and (left[..., 1] == 1.0).all() and (left[..., 3] == 0.25).all()
    assert (left[..., 2] == np.float32(3 + 12 + 27)/np.float32(32)).all()        # row 2 of iMix times (1, 2, 3)
    assert (right == 0).all()                                                     # discarded inside keep()
    N.check(gpu.lib.sfx_program_destroy(prog))


def test_derivatives_and_the_fused_kernels(gpu):
    """dFdx/dFdy/fwidth need 2x2 neighbours in the lanes of a quad: the unfused kernel lays them out so, the fused kernel has that
    layout at ssaa 2 only — the program says so, the other factors are refused, and scenes render either way with the same bytes
    from tape and frame loop"""
    from shaderflow_amd import ShaderScene
    text = (HERE/"golden"/"jit"/"edges.glsl").read_text()
    prog, _ = load(gpu, text)
    assert [gpu.lib.sfx_program_fusable(prog, ssaa) for ssaa in (1, 2, 4)] == [0, 1, 0]
    gpu.set_uniforms(prog, O.default_uniforms(64, 36, iSSAA=2.0))
    final = gpu.empty(64, 36, 3)
    assert gpu.lib.sfx_render_resolve(prog, final, 1, 1) == N.E_UNSUPPORTED and gpu.lib.sfx_render_resolve(prog, final, 4, 2) == N.E_UNSUPPORTED
    fused = gpu.render_resolve(prog, 64, 36, 2, 2)                               # ssaa 2: the supersamples of a pixel are a quad
    two_pass = gpu.resolve(gpu.render(prog, 128, 72), 64, 36, 2)
    assert np.abs(fused.astype(int) - two_pass.astype(int)).max() <= 1 and fused.std() > 10
    N.check(gpu.lib.sfx_program_destroy(prog))
    plain, _ = load(gpu, GRADIENT)
    assert [gpu.lib.sfx_program_fusable(plain, ssaa) for ssaa in (1, 2, 4)] == [1, 1, 1]
    N.check(gpu.lib.sfx_program_destroy(plain))

    class Edges(ShaderScene):
        def build(self):
            super().build()
            self.shader.fragment = text

    for ssaa in (1, 2):                                                          # two passes, fused
        kw = dict(width=96, height=54, fps=30, time=3/30, ssaa=ssaa, output=bytes)
        loop = np.frombuffer(Edges().main(batch=False, **kw), np.uint8).reshape(3, 54, 96, 3)
        tape = np.frombuffer(Edges().main(batch=None, **kw), np.uint8).reshape(3, 54, 96, 3)
        assert np.array_equal(loop, tape) and loop.std() > 10
    # against the golden at the scene's render resolution would need the resolve; the ring is there and anti-aliased:
    centre_row = loop[0, 27].astype(int)
    assert len(np.unique(centre_row[:, 0])) > 4


@pytest.mark.parametrize("name", ["waves", "march", "cells", "hash", "builtins", "materials", "polar"])
def test_device_and_host_builds_of_a_translation_agree_bit_for_bit(gpu, name):
    """One translation unit, two targets: the gfx950 code object and the x86 host build of the same header (tests/jit_host.py). Every
    operation is a correctly rounded binary32 operation on both (no contraction, IEEE division and square root, denormals kept), so the
    float outputs must be identical — this ties the GPU to the CPU checks of tests/test_host_translate.py (oracle, OpenGL goldens)."""
    from tests.jit_host import HostFragment
    case = CASES[name]
    w, h = case["width"], case["height"]
    text = (HERE/"golden"/"jit"/f"{name}.glsl").read_text()
    prog, translation = load(gpu, text, [("sampler2D", "background")])
    host = HostFragment(translation, CACHE)
    overrides = {k: (tuple(v) if isinstance(v, list) else v) for k, v in case["uniforms"].items()}
    u = O.default_uniforms(w, h, iTime=1.375, iCameraZoom=0.9, **{k: v for k, v in overrides.items() if k not in ("iTime", "iCameraZoom")})
    gpu.set_uniforms(prog, u)
    host.set_uniforms(u)
    for key, value in case["floats"].items():
        gpu.set_values(prog, key, value); host.set(key, value)
    for key, value in case["integers"].items():
        gpu.set_values(prog, key, value, integer=True); host.set(key, value)
    gpu.bind(prog, "background", gpu.texture(G["background"], "linear", True, True))
    host.bind("background", G["background"], "linear", True, True)
    got = gpu.render(prog, w, h, comps=4, dtype=np.float32)
    want = host.render_float(w, h)
    N.check(gpu.lib.sfx_program_destroy(prog))
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), f"{int((~same).sum())} of {same.size} floats differ; first at {np.argwhere(~same)[0].tolist()}: {got[~same][0]!r} vs {want[~same][0]!r}"


# ---- the LDS tile of tap-heavy fragments (jit_runtime.hpp TileView / JitShader<…, true>) ---------------------------------------
BLUR = """
uniform float radius = 3.0;
uniform vec2 drift = vec2(0.0);
uniform float swirl = 0.0;
void main() {
    vec4 sum = vec4(0.0);
    float total = 0.0;
    vec2 texel = 1.0/vec2(textureSize(background, 0));
    vec2 centre = rotate2d(swirl*length(agluv))*(astuv - 0.5) + 0.5 + drift;
    for (int x = -4; x <= 4; x++) {
        for (int y = -4; y <= 4; y++) {
            float weight = exp(-float(x*x + y*y)/(radius*radius));
            sum += weight*texture(background, centre + vec2(x, y)*texel*1.37);
            total += weight;
        }
    }
    fragColor = sum/total;
}
"""


def _tile_texture(kind: str, rng) -> np.ndarray:
    if kind == "rgba8":
        return rng.integers(0, 256, (54, 96, 4), dtype=np.uint8)
    if kind == "rgb8":
        return rng.integers(0, 256, (37, 61, 3), dtype=np.uint8)
    if kind == "wide":                                   # a block's footprint is larger than the tile: the crop and the generic fetch share the taps
        return rng.integers(0, 256, (360, 640, 4), dtype=np.uint8)
    return rng.random((40, 70, 2), dtype=np.float32)      # "rg32f"


@pytest.mark.parametrize("kind,filter,repeat,drift,swirl", [
    ("rgba8", "linear", True, (0.0, 0.0), 0.0),
    ("rgba8", "linear", False, (-0.31, 0.93), 0.0),      # taps left of and above the texture: clamped copies in the tile
    ("rgba8", "nearest", True, (0.6, -1.2), 0.0),        # wrapped several times over
    ("rgb8", "linear", True, (0.013, 0.007), 0.0),
    ("rg32f", "linear", False, (0.0, 0.0), 0.0),
    ("wide", "linear", True, (0.0, 0.0), 0.0),
    ("rgba8", "linear", True, (0.0, 0.0), 2.5),          # not monotone across a block: taps of the middle pixels leave the probed box
    ("wide", "nearest", False, (0.2, 0.1), 1.0),
])
def test_tiled_sampler_gives_the_bits_of_the_host_build(gpu, kind, filter, repeat, drift, swirl):
    """The tile is a cache: whatever was staged, a tap blends the same decoded texels with the same operations. The host build of the
    same translation has no tile at all, so float-for-float equality checks the staged values, the wrap at stage time and the bounds."""
    from tests.jit_host import HostFragment
    prog, translation = load(gpu, BLUR, [("sampler2D", "background")])
    assert translation.tiled_sampler == "background"
    host = HostFragment(translation, CACHE)
    data = _tile_texture(kind, np.random.default_rng(5))
    w, h = 150, 43                                        # partial blocks on both axes
    u = O.default_uniforms(w, h, iTime=0.5)
    gpu.set_uniforms(prog, u); host.set_uniforms(u)
    for key, value in (("radius", 2.5), ("drift", drift), ("swirl", swirl)):
        assert gpu.set_values(prog, key, value); host.set(key, value)
    gpu.bind(prog, "background", gpu.texture(data, filter, repeat, repeat))
    host.bind("background", data, filter, repeat, repeat)
    got = gpu.render(prog, w, h, comps=4, dtype=np.float32)
    want = host.render_float(w, h)
    N.check(gpu.lib.sfx_program_destroy(prog))
    same = got.view(np.uint32) == want.view(np.uint32)
    assert same.all(), f"{int((~same).sum())} of {same.size} floats differ; first at {np.argwhere(~same)[0].tolist()}"
    assert np.isfinite(got).all() and got[..., 0].std() > 0.01         # a picture, not NaNs agreeing with NaNs


def test_tiled_and_untiled_translations_write_the_same_frames(gpu, monkeypatch):
    """Every kernel of the code object (plain render, fused 1x / 2x / 4x) with the tile against the same fragment translated without it"""
    data = _tile_texture("rgba8", np.random.default_rng(9))
    frames = {}
    for tile in ("1", "0"):
        monkeypatch.setenv("SHADERFLOW_JIT_TILE", tile)
        prog, translation = load(gpu, BLUR, [("sampler2D", "background")])
        assert (translation.tiled_sampler is not None) == (tile == "1")
        gpu.bind(prog, "background", gpu.texture(data, "linear", True, False))
        for (ssaa, subsample) in ((1, 1), (2, 2), (2, 1), (4, 4), (4, 2)):
            w, h = 301, 37
            gpu.set_uniforms(prog, O.default_uniforms(w, h, iTime=0.4, iSSAA=float(ssaa)))
            assert gpu.set_values(prog, "radius", 2.0) and gpu.set_values(prog, "drift", (0.4, -0.2)) and gpu.set_values(prog, "swirl", 0.3)
            frames[tile, ssaa, subsample] = gpu.render_resolve(prog, w, h, ssaa, subsample)
            assert frames[tile, ssaa, subsample].std() > 5          # a picture, not a blank frame
        frames[tile, "render"] = gpu.render(prog, 131, 67)
        N.check(gpu.lib.sfx_program_destroy(prog))
    for key in [k[1:] for k in frames if k[0] == "1"]:
        assert np.array_equal(frames[("1",) + key], frames[("0",) + key]), key


def test_a_fragment_that_never_taps_the_tiled_sampler_still_renders(gpu, monkeypatch):
    """Forced tile on a sampler the probes do not read (the branch is off): an empty box, nothing staged, every later tap generic"""
    monkeypatch.setenv("SHADERFLOW_JIT_TILE", "background")
    text = """
    uniform float gate = 0.0;
    void main() {
        fragColor = vec4(stuv, 0.25, 1.0);
        if (gate > 0.5 && astuv.x > 0.3 && astuv.x < 0.6) fragColor = texture(background, astuv);
    }
    """
    prog, translation = load(gpu, text, [("sampler2D", "background")])
    assert translation.tiled_sampler == "background"
    data = _tile_texture("rgba8", np.random.default_rng(2))
    gpu.bind(prog, "background", gpu.texture(data, "nearest", False, False))
    w, h = 192, 16
    gpu.set_uniforms(prog, O.default_uniforms(w, h))
    gpu.set_values(prog, "gate", 0.0)
    off = gpu.render(prog, w, h)
    gpu.set_values(prog, "gate", 1.0)
    on = gpu.render(prog, w, h)
    N.check(gpu.lib.sfx_program_destroy(prog))
    columns = (np.arange(w) + 0.5)/w
    inside = (columns > 0.3) & (columns < 0.6)
    assert np.array_equal(on[:, ~inside], off[:, ~inside])
    ix = np.floor(columns*96).astype(int)
    iy = np.floor((np.arange(h) + 0.5)/h*54).astype(int)
    assert np.array_equal(on[:, inside], data[iy][:, ix[inside]])


def test_tiled_sampler_with_taps_far_away_on_both_sides(gpu, monkeypatch):
    """A probe whose taps span the whole clamped coordinate range (+-2^24 texels each way): the box is cropped to the tile's capacity
    about its centre without overflowing, and every tap — near ones, far ones outside the crop — equals the untiled translation's
    (the host build is no reference here: float -> int conversion of such coordinates saturates on the device and does not on x86)"""
    text = """
    uniform float far = 1.0e9;
    void main() {
        vec4 sum = vec4(0.0);
        for (int k = -3; k <= 3; k++) sum += texture(background, astuv*0.05 + vec2(k, -k)/96.0);
        sum += texture(background, vec2(far, far)) + texture(background, vec2(-far, -far));
        fragColor = vec4(sum.rgb/9.0, texture(background, vec2(far, -far)*1.0e30).a);      // infinite coordinates: a NaN, kept out of rgb
    }
    """
    data = _tile_texture("rgba8", np.random.default_rng(11))
    w, h = 130, 21
    frames = {}
    for tile in ("1", "0"):
        monkeypatch.setenv("SHADERFLOW_JIT_TILE", tile)
        prog, translation = load(gpu, text, [("sampler2D", "background")])
        assert (translation.tiled_sampler == "background") == (tile == "1")
        gpu.set_uniforms(prog, O.default_uniforms(w, h))
        assert gpu.set_values(prog, "far", 1.0e9)                  # (sfx_program_load knows no initialisers: ShaderProgram applies them)
        for repeat in (True, False):
            gpu.bind(prog, "background", gpu.texture(data, "linear", repeat, repeat))
            frames[tile, repeat] = gpu.render(prog, w, h, comps=4, dtype=np.float32)
        N.check(gpu.lib.sfx_program_destroy(prog))
    for repeat in (True, False):
        assert np.array_equal(frames["1", repeat].view(np.uint32), frames["0", repeat].view(np.uint32)), repeat
        assert np.isfinite(frames["1", repeat][..., :3]).all() and frames["1", repeat][..., :3].std() > 0.01


def test_tiled_sampler_passed_to_a_helper_function(gpu):
    """`vec4 blur(sampler2D tex, …)` called with the tiled sampler: the view travels inside the sampler value, the probe's record too"""
    from tests.jit_host import HostFragment
    text = """
    vec4 blur(sampler2D tex, vec2 uv, vec2 step) {
        vec4 s = vec4(0.0);
        for (int k = -5; k <= 5; k++) s += texture(tex, uv + float(k)*step);
        return s/11.0;
    }
    void main() {
        vec2 texel = 1.0/vec2(textureSize(background, 0));
        fragColor = 0.5*blur(background, astuv, vec2(texel.x, 0.0)) + 0.5*blur(background, astuv, vec2(0.0, texel.y));
    }
    """
    prog, translation = load(gpu, text, [("sampler2D", "background")])
    assert translation.tiled_sampler == "background"
    host = HostFragment(translation, CACHE)
    data = _tile_texture("rgb8", np.random.default_rng(21))
    w, h = 97, 41
    u = O.default_uniforms(w, h)
    gpu.set_uniforms(prog, u); host.set_uniforms(u)
    gpu.bind(prog, "background", gpu.texture(data, "linear", False, True))
    host.bind("background", data, "linear", False, True)
    got = gpu.render(prog, w, h, comps=4, dtype=np.float32)
    want = host.render_float(w, h)
    N.check(gpu.lib.sfx_program_destroy(prog))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
