"""
The oracle against THE REFERENCE ITSELF: tests/golden/mesa.npz holds frames that /root/reference's own Python (ShaderScene.main and
everything under it, unmodified) rendered in the build container through a real desktop OpenGL — Mesa llvmpipe 4.5 core, the software
rasteriser the north star names — with its GLSL as shader.py:190-239 assembles it (`#version 330`, typed uniforms). See
tests/golden/make_golden_mesa.py, refhost.py and mesa_shim.c for how. This is what pins the pixel half of the oracle (and, through
the scene exports, the audio half once more, end to end); gles.npz (SwiftShader, rewritten GLSL ES) stays as a second witness in
tests/test_oracle_gles.py. CPU only.

Bounds. `max ≤ 1 LSB` wherever the measured histogram says so — most images. Where a fragment reads an 8-bit texture through the
bilinear filter twice (a textured fragment at ssaa 1 followed by final.glsl's tent, or multipass' second layer), llvmpipe's filter —
8 fractional bits of weight for unorm8 textures, an implementation choice OpenGL allows (≥ 4 subtexel bits) — puts up to 1.3 % of
the values 2 LSB away, never 3; SwiftShader's images of the same inputs sit with the oracle there (tools/parity_histogram.py).
Chaotic fragments (tetration's boundary, default.glsl's 1/circle² ring) are held to a fraction, as in the older set.
"""
from pathlib import Path

import numpy as np
import pytest

from oracle import binding as O
from shaderflow_amd import synth
from tests import replay as R
from tests.helpers import i16_to_f32, oracle_textures, visualizer_inputs

G = np.load(Path(__file__).parent/"golden"/"mesa.npz")
CAMERAS = {"plain": {}, "moved": dict(iCameraZoom=1.3, iCameraIsometric=0.2, iCameraPosition=(0.1, -0.05, 0.0)),
           "stereo": dict(iCameraProjection=1, iCameraSeparation=0.07, iCameraZoom=1.2), "equirect": dict(iCameraProjection=2, iCameraZoom=0.8)}


def agree(tag: str, want: np.ndarray, fraction: float = 1.0, bound: int = 1, key: str = "image") -> None:
    got = G[f"{tag}.{key}"]
    want = want[..., :got.shape[2]]
    assert got.shape == want.shape, (tag, got.shape, want.shape)
    d = np.abs(got.astype(int) - want.astype(int))
    within = (d <= 1).mean()
    assert within >= fraction, f"{tag}.{key}: {100*within:.3f}% within 1 LSB (max {d.max()})"
    if fraction == 1.0 or bound > 1:
        assert d.max() <= bound, f"{tag}.{key}: max {d.max()} LSB"
    assert np.abs(got.astype(float).mean() - want.astype(float).mean()) < 0.25, tag       # no systematic offset


def test_the_fixture_was_rendered_by_a_desktop_opengl():
    assert "llvmpipe" in str(G["meta.renderer"]) and "Core Profile" in str(G["meta.renderer"])


@pytest.mark.parametrize("camera", list(CAMERAS))
def test_default_fragment_every_projection(camera):
    u = O.default_uniforms(160, 90, iTime=0.75, iTau=0.3, **CAMERAS[camera])
    screen = O.render("default", u, {}, 160, 90, threads=4)
    fraction = {"plain": 0.996, "stereo": 0.999}.get(camera, 1.0)          # the ring: 1/circle² next to circle = 0
    agree(f"default.{camera}", screen, fraction)
    agree(f"default.{camera}", O.resolve(screen, 160, 90, 2), fraction, key="final")


def test_untextured_fragments():
    u = O.default_uniforms(96, 54, iTime=3.0, iTau=0.3)
    got = G["shadertoy.image"]
    assert np.array_equal(got, O.render("shadertoy", u, {}, 96, 54))
    assert np.array_equal(G["multi_child.image"], O.render("multi_child", O.default_uniforms(64, 36), {}, 64, 36))
    assert np.array_equal(G["raymarch.image"], O.render("raymarch", O.default_uniforms(160, 90), {}, 160, 90, threads=4))
    assert np.array_equal(G["raymarch.moved.image"],
                          O.render("raymarch", O.default_uniforms(160, 90, iCameraPosition=(0.4, 0.2, -1.5), iCameraZoom=0.8), {}, 160, 90, threads=4))
    agree("mandelbrot", O.render("mandelbrot", O.default_uniforms(160, 90, iQuality=0.2), {}, 160, 90, threads=4), 0.9999, bound=2)


def test_missing_fragment_reads_an_uninitialised_output():
    """fragment/missing.glsl:14-18 ACCUMULATES into `fragColor`, which nothing initialised: undefined in GLSL. Drivers that start
    outputs at zero (SwiftShader, the vendors' desktop drivers) draw the magenta checkerboard, and that is the convention of the
    oracle and of the kernels (gles.npz pins it); Mesa's compiler treats the read as undefined and llvmpipe's image is neither
    checkerboard nor stable. What both agree on is the alpha the shader does assign. Kept as a witness of why this fragment is
    pinned on the other implementation."""
    image = G["missing.image"]
    assert (image[..., 3] == 51).all()                                         # 0.2 → 51
    want = O.render("missing", O.default_uniforms(96, 54, iTime=3.0, iTau=0.3), {}, 96, 54)
    assert (want[..., 3] == 51).all() and not np.array_equal(image, want)


@pytest.mark.parametrize("volume", [0.0, 0.5, 1.2])
def test_visualizer_with_its_radial_blur(volume):
    """The benchmark fragment on a pure-noise background (the worst case for a filter's weight precision). At volume 0 all 91 taps
    coincide, so llvmpipe's 8-bit weights are not averaged out: 30 of 57 600 values are 2 LSB off, SwiftShader agrees with the oracle"""
    u, arrays, params = visualizer_inputs(160, 90, seed=21, volume=volume, bg_size=(120, 68))
    screen = O.render("visualizer", u, oracle_textures(arrays, params), 160, 90, threads=8)
    if volume == 0.0:
        agree(f"visualizer.v{volume}", screen, 0.999, bound=2)
    else:
        agree(f"visualizer.v{volume}", screen)


def test_visualizer_supersampled_and_resolved():
    """BASELINE config 3 in small: visualizer.frag at 2x SSAA, then final.glsl"""
    u, arrays, params = visualizer_inputs(192, 108, seed=33, volume=0.9, bg_size=(160, 90))
    u.iSSAA = 2.0
    screen = O.render("visualizer", u, oracle_textures(arrays, params), 384, 216, threads=8)
    agree("visualizer.ssaa2", screen)
    agree("visualizer.ssaa2", O.resolve(screen, 192, 108, 2, threads=4), key="final")


def test_audio_texture_fragments():
    u, arrays, params = visualizer_inputs(128, 72, seed=5)
    arrays["iSpectrogram"] = arrays["iSpectrogram"]*3
    for name in ("bars", "waveform"):
        assert np.array_equal(G[f"{name}.image"], O.render(name, u, oracle_textures(arrays, params), 128, 72, threads=4)), name
    u.user[0] = 0.35
    agree("dynamics", O.render("dynamics", u, oracle_textures(arrays, params), 128, 72))


@pytest.mark.parametrize("filter", ["nearest", "linear"])
@pytest.mark.parametrize("wrap", ["clamp", "repeat"])
def test_sampler_addressing_and_filtering(filter, wrap):
    """texture() on a 7x5 RGBA8 grid at coordinates from -0.75 to 1.75: texel addressing, wrap modes, bilinear weights"""
    texture = O.make_texture(G["sampler.texels"], filter, wrap == "repeat", wrap == "repeat")
    w, h = 70, 50
    want = np.zeros((h, w, 4), np.uint8)
    one, half, two = np.float32(1), np.float32(0.5), np.float32(2)
    for j in range(h):
        for i in range(w):
            s = np.float32(((np.float32(i) + half)/np.float32(w)*two - one + one)/two)          # vertex/default.glsl:9-10
            t = np.float32(((np.float32(j) + half)/np.float32(h)*two - one + one)/two)
            c = O.sample(texture, s*np.float32(2.5) - np.float32(0.75), t*np.float32(2.5) - np.float32(0.75))
            want[j, i] = np.rint(np.clip(c, 0, 1)*255)
    if filter == "nearest":
        assert np.array_equal(G[f"sampler.{filter}.{wrap}.image"], want)                        # no filtering arithmetic: identical
    else:
        agree(f"sampler.{filter}.{wrap}", want, 0.99, bound=2)                                   # random texels: the 8-bit weights show


def test_final_glsl_every_kernel():
    screen = G["final.screen"]
    for (fw, fh, sub) in ((64, 36, 2), (64, 36, 1), (128, 72, 2), (32, 18, 4)):
        want = O.resolve(screen, fw, fh, sub)
        got = G[f"final.{fw}x{fh}.k{sub}.image"]
        d = np.abs(got[..., :3].astype(int) - want.astype(int))
        assert d.max() <= (1 if fw*sub == 128 or sub == 1 else 2), (fw, fh, sub, d.max())       # taps on texel centres: ≤ 1; between: the filter's bits
        assert (d <= 1).mean() >= 0.99, (fw, fh, sub)


@pytest.mark.parametrize("tag,kw", [("tetration", {}), ("tetration.zoomed", dict(iCameraZoom=2.5, iCameraPosition=(-0.7, 0.1, 0.0)))])
def test_tetration_integer_division(tag, kw):
    """`it / MAX_STEPS` is an integer division — here compiled as one, the GLSL is not rewritten. The iteration is chaotic along the
    fractal's boundary, where the built-ins' last bits decide between escape and hue: ≥ 99.5 % of the values within 1 LSB"""
    agree(tag, O.render("tetration", O.default_uniforms(160, 90, **kw), {}, 160, 90, threads=4), fraction=0.995)


# ---- the reference's example scenes, exported by scene.main() ---------------------------------------------------------------------

def scene(tag: str, want: np.ndarray, fraction: float = 1.0, bound: int = 1) -> None:
    got = G[f"scene.{tag}.frames"]
    assert got.shape == want.shape, (tag, got.shape, want.shape)
    for n, k in enumerate(G[f"scene.{tag}.index"]):
        d = np.abs(got[n].astype(int) - want[n].astype(int))
        assert d.max() <= bound and (d <= 1).mean() >= fraction, f"scene.{tag} frame {k}: max {d.max()}, {100*(d <= 1).mean():.3f}% within 1"


def test_basic_scene_baseline_config_1():
    scene("basic", R.plain_scene("default", 256, 256, 1, 2, 60.0, 6, pick=(0, 5)))


def test_clock_only_scenes():
    scene("shadertoy", R.plain_scene("shadertoy", 96, 54, 1, 2, 60.0, 3, pick=(2,)))
    scene("raymarch", R.plain_scene("raymarch", 96, 54, 2, 2, 60.0, 3, pick=(2,)))
    scene("multishader", R.multishader_scene(64, 36, 60.0, 2, (1,)))


def test_layered_and_temporal_scenes():
    street = synth.background_image(480, 270)
    scene("multipass", R.multipass_scene(street, 128, 72, 1, 60.0, 3), 0.985, bound=2)          # two chained bilinear fetches of 8-bit textures
    frames = R.motionblur_scene(street, 96, 54, 60.0, 14)
    scene("motionblur", frames[[0, 1, 8, 9, 10, 13]])
    assert not G["scene.motionblur.frames"][:3].any() and G["scene.motionblur.frames"][3].any()     # black until temporal-1 frames have been rendered
    scene("life", R.life_scene(G["scene.life.first"], 128, 72, 60.0, 20)[[0, 1, 5, 6, 7, 12, 13, 19]])


def reference_uniforms(tag: str, frame: int, w: int, h: int) -> "O.Uniforms":
    """The values the REFERENCE's pipeline emitted for that frame (captured by a module of the exported scene), as oracle uniforms"""
    names, sizes, values = G[f"scene.{tag}.uniform_names"], G[f"scene.{tag}.uniform_sizes"], G[f"scene.{tag}.uniforms"][frame]
    u, at = O.default_uniforms(w, h), 0
    for name, size in zip(names, sizes):
        value = values[at:at + size]
        at += size
        if hasattr(u, str(name)):
            current = getattr(u, str(name))
            if hasattr(current, "__len__"):
                for i in range(len(current)):
                    current[i] = float(value[i])
            else:
                setattr(u, str(name), type(current)(value[0]))
    return u


def test_scene_that_drives_the_camera():
    """move / zoom / rotate2d / projection switch from update(): the reference's camera systems (float64 DynamicNumbers, quaternion
    rotation → right/up/forward) produce the uniforms, its camera.glsl the pixels; the oracle on those uniforms gives the frames"""
    got = G["scene.moving_camera.frames"]
    for k in range(6):
        u = reference_uniforms("moving_camera", k, 96, 54)
        want = O.resolve(O.render("default", u, {}, 96, 54, threads=4), 96, 54, 2)
        d = np.abs(got[k].astype(int) - want.astype(int))
        assert (d <= 1).mean() >= 0.99, (k, d.max(), (d <= 1).mean())           # default.glsl's ring: a 96 x 54 frame has little else
    assert not np.array_equal(got[0], got[5])
    names = list(G["scene.moving_camera.uniform_names"])
    at = int(np.sum(G["scene.moving_camera.uniform_sizes"][:names.index("iCameraProjection")]))
    assert G["scene.moving_camera.uniforms"][2][at] == 0 and G["scene.moving_camera.uniforms"][5][at] == 1


def test_scene_with_python_logic_between_frames():
    scene("dynamics", R.dynamics_scene(synth.background_image(480, 270), 128, 72, 60.0, 90, (0, 1, 30, 59, 61, 89)), 0.99, bound=2)


@pytest.mark.parametrize("smooth", [False, True])
def test_scrolling_spectrogram_scene(smooth):
    """ShaderSpectrogram(length = 0.5 s) on the reference: a 30-column texture scrolled by its numpy code and shown by a fragment of
    this repository's own, compiled by Mesa. Here: the oracle's audio tape replays spectrogram.py:298-311 and the HOST build of the
    run-time translation of the same text (tests/jit_host.py) shades the frames — within 1 LSB"""
    from shaderflow_amd import glsl2hip
    from tests.helpers import SCROLL_FRAGMENT
    from tests.jit_host import HostFragment
    tag = f"scroller.{'smooth' if smooth else 'nearest'}"
    w, h, ssaa, _, fps, frames = G[f"scene.{tag}.args"]
    w, h, ssaa, frames = int(w), int(h), int(ssaa), int(frames)
    pcm, samplerate, width = synth.sweep_clip(2.0, 44100), 44100, int(0.5*fps)
    planar = np.ascontiguousarray(pcm.T)
    times, dts, rdts = O.clock(fps, frames)
    _, tell = O.reader(rdts, samplerate, 2, planar.shape[1])
    fmin, fmax, bins = O.from_notes(O.lib().sfo_note_of_frequency(20.0, 440.0), O.lib().sfo_note_of_frequency(14000.0, 440.0), True)
    indptr, indices, data = O.filterbank(0, 0, fmin, fmax, bins, 12, samplerate)
    spec = O.DynF32(2*bins, 4, 1, 0)
    texture = np.zeros((bins, width, 2), np.float32)
    host = HostFragment(glsl2hip.translate(SCROLL_FRAGMENT, [("sampler2D", "iSpectrogram")]), Path(__file__).parent.parent/"build"/"jit")
    keep = {int(k): n for n, k in enumerate(G[f"scene.{tag}.index"])}
    offset = 0
    for k in range(frames):
        offset = (offset + 1) % width
        target = O.csr_dot(indptr, indices, data, O.fft_power(planar, int(tell[k])))
        texture[:, offset, :] = spec.step(target.ravel(), abs(dts[k])).reshape(bins, 2)
        if k not in keep:
            continue
        u = O.default_uniforms(w, h, iTime=times[k], iTau=(times[k]/(frames/fps)) % 1.0, iDuration=frames/fps, iDeltatime=dts[k], iFramerate=fps,
                               iFrame=round(times[k]*fps), iSSAA=float(ssaa), iSubsample=2, iSpectrogramLength=width, iSpectrogramBins=bins,
                               iSpectrogramSmooth=int(smooth), iSpectrogramOffset=offset/width)
        host.set_uniforms(u)
        host.bind("iSpectrogram", texture.copy(), "linear" if smooth else "nearest", True, False)
        want = O.resolve(host.render(w*ssaa, h*ssaa), w, h, 2)
        d = np.abs(G[f"scene.{tag}.frames"][keep[k]].astype(int) - want.astype(int))
        assert d.max() <= 1, (tag, k, d.max(), (d > 1).sum())


def test_audio_scenes_end_to_end():
    """Audio file → the reference's numpy STFT, filterbank, DynamicNumbers, waveform → its GLSL on llvmpipe → final.glsl → the encoder
    pipe, against the oracle's audio tape + fragments on the same clip: the north star's parity statement, reference on one side"""
    P = np.load(Path(__file__).parent/"golden"/"pipeline.npz")
    fps, samplerate, frames = float(P["meta"][0]), int(P["meta"][1]), int(P["meta"][2])
    pcm, background = i16_to_f32(P["pcm_i16"]), synth.background_image(240, 135, seed=7)
    scene("visualizer", R.audio_scene("visualizer", pcm, samplerate, background, 192, 108, 2, 2, fps, frames, pick=(0, 1, 10, 40, 99, frames - 1)), 0.9995, bound=2)
    scene("visualizer.ssaa1", R.audio_scene("visualizer", pcm, samplerate, background, 192, 108, 1, 2, fps, 60, pick=(1, 30, 59)), 0.99, bound=2)
    scene("musicbars", R.audio_scene("bars", pcm, samplerate, None, 160, 90, 2, 2, fps, 60, pick=(1, 30, 59), high=18000.0))
    want = R.audio_scene("waveform", pcm, samplerate, None, 160, 90, 2, 2, fps, 60, pick=(1, 30, 59), waveform_smooth=False)
    assert np.array_equal(G["scene.waveform.frames"], want)


# ---- what llvmpipe's filter is, and what it explains -----------------------------------------------------------

F = np.load(Path(__file__).parent/"golden"/"filter.npz")


@pytest.mark.parametrize("tag", ["row", "grid.repeat", "grid.clamp"])
def test_llvmpipe_filters_unorm8_textures_in_8_bit_fixed_point(tag):
    """filter.npz: `texture(probe, astuv*S + O)` rendered by the reference on llvmpipe into a FLOAT32 target. Every filtered value is
    k/255 — the filter's result is rounded back to 8 bits — and the oracle's llvmpipe switch (24.8 fixed-point coordinates, 8-bit
    weights, a + ((w·(b − a) + 128) >> 8) along x then y) reproduces all of them, bit for bit; the float-weight filter of OpenGL's
    specification does not (by at most 1 LSB: the source of every 2 LSB once a second filter or a quantisation follows)."""
    assert float(F["filter.max_distance_from_k_over_255"]) < 1e-6           # the fixture's float32 values sit ON the 8-bit grid
    texels = F[f"filter.{tag.split('.')[0]}.texels"]
    width, height, sx, sy, ox, oy, repeat = F[f"filter.{tag}.args"]
    width, height, repeat = int(width), int(height), bool(repeat)
    texture = O.make_texture(texels, "linear", repeat, repeat)
    want = F[f"filter.{tag}.k"]
    one, half = np.float32(1), np.float32(0.5)
    rows = range(height) if height <= 2 else range(0, height, 7)              # every 7th row of the grids keeps this under a second
    model, spec = np.zeros((len(rows), width, 4), np.uint8), np.zeros((len(rows), width, 4), np.uint8)
    for n, j in enumerate(rows):
        t = (np.float32(j) + half)/np.float32(height)*np.float32(sy) + np.float32(oy)
        for i in range(width):
            s = (np.float32(i) + half)/np.float32(width)*np.float32(sx) + np.float32(ox)
            with O.llvmpipe_filter():
                model[n, i] = np.rint(O.sample(texture, s, t)*255)
            spec[n, i] = np.rint(np.clip(O.sample(texture, s, t), 0, 1)*255)
    kept = want[list(rows)]
    assert np.array_equal(model, kept), np.argwhere(model != kept)[:4]
    d = np.abs(spec.astype(int) - kept.astype(int))
    assert d.max() == 1 and 0.02 < (d == 1).mean() < 0.5                       # the specification's filter: close, not equal


@pytest.mark.parametrize("tag", ["row", "grid.repeat", "grid.clamp"])
def test_the_products_fixed_point_filter_is_llvmpipes(tag):
    """The PRODUCT's opt-in filter model (csrc/glsl.hpp texture_fixed8, `sfx_ctx_filter_model`) is the same arithmetic in the kernels'
    own header: the HOST build of the run-time translation of the probe's text (tests/jit_host.py compiles csrc/jit_runtime.hpp for the
    CPU) reproduces every value llvmpipe filtered, byte for byte. The device build of the same header: tests/test_gpu_filter_model.py."""
    import ctypes as C

    from shaderflow_amd import glsl2hip
    from tests.jit_host import HostFragment
    texels = np.ascontiguousarray(F[f"filter.{tag.split('.')[0]}.texels"])
    width, height, sx, sy, ox, oy, repeat = F[f"filter.{tag}.args"]
    width, height, repeat = int(width), int(height), bool(repeat)
    text = f"void main() {{ fragColor = texture(probe, astuv*vec2({float(sx)!r}, {float(sy)!r}) + vec2({float(ox)!r}, {float(oy)!r})); }}"
    host = HostFragment(glsl2hip.translate(text, [("sampler2D", "probe")]), Path(__file__).parent.parent/"build"/"jit")
    host.set_uniforms(O.default_uniforms(width, height))
    binding = next(b for b in host.translation.bindings if b.name == "probe")
    FILTER_LINEAR_FIXED8 = 4                                                         # csrc/glsl.hpp
    host.lib.sfx_jit_host_texture(host.textures, binding.slot, texels.ctypes.data_as(C.c_void_p), texels.shape[1], texels.shape[0], texels.shape[2],
                                  0, FILTER_LINEAR_FIXED8, int(repeat), int(repeat))
    assert np.array_equal(host.render(width, height), F[f"filter.{tag}.k"])


LLVMPIPE_CASES = {
    # tag: (oracle image under the switch, exact?)  — the bounds of the tests above WITHOUT the switch: max 2, up to 1.3 % at 2
    "sampler.linear.clamp": True, "sampler.linear.repeat": True, "final.64x36.k1": True, "dynamics": True,
    "visualizer.v0.0": False, "visualizer.v0.5": False, "visualizer.v1.2": False,
    "final.64x36.k2": False, "final.128x72.k2": False, "final.32x18.k4": False,
}


def _probe_under_the_switch(tag: str) -> tuple[np.ndarray, np.ndarray]:
    if tag.startswith("sampler"):
        wrap = tag.split(".")[2]
        texture = O.make_texture(G["sampler.texels"], "linear", wrap == "repeat", wrap == "repeat")
        w, h = 70, 50
        want = np.zeros((h, w, 4), np.uint8)
        one, half, two = np.float32(1), np.float32(0.5), np.float32(2)
        for j in range(h):
            for i in range(w):
                s = np.float32(((np.float32(i) + half)/np.float32(w)*two - one + one)/two)
                t = np.float32(((np.float32(j) + half)/np.float32(h)*two - one + one)/two)
                want[j, i] = np.rint(np.clip(O.sample(texture, s*np.float32(2.5) - np.float32(0.75), t*np.float32(2.5) - np.float32(0.75)), 0, 1)*255)
        return G[f"{tag}.image"], want
    if tag.startswith("final"):
        size, k = tag.split(".")[1], int(tag.split(".")[2][1:])
        fw, fh = (int(v) for v in size.split("x"))
        return G[f"{tag}.image"][..., :3], O.resolve(G["final.screen"], fw, fh, k)
    if tag == "dynamics":
        u, arrays, params = visualizer_inputs(128, 72, seed=5)
        u.user[0] = 0.35
        return G["dynamics.image"], O.render("dynamics", u, oracle_textures(arrays, params), 128, 72)
    volume = float(tag[len("visualizer.v"):])
    u, arrays, params = visualizer_inputs(160, 90, seed=21, volume=volume, bg_size=(120, 68))
    return G[f"{tag}.image"], O.render("visualizer", u, oracle_textures(arrays, params), 160, 90, threads=8)


@pytest.mark.parametrize("tag", list(LLVMPIPE_CASES))
def test_probes_beyond_one_lsb_are_the_filters_weights(tag):
    """Every probe the tests above hold to `max ≤ 2`: with the oracle filtering unorm8 textures as llvmpipe does, the same comparison
    gives max ≤ 1 — byte-identical for the sampler, the 1:1 resolve and the textured inline fragment. The cause of the 2-LSB values
    is demonstrated, not asserted; the kernels keep the specification's float weights (the switch is the checker's)."""
    with O.llvmpipe_filter():
        got, want = _probe_under_the_switch(tag)
    want = want[..., :got.shape[2]]
    d = np.abs(got.astype(int) - want.astype(int))
    if LLVMPIPE_CASES[tag]:
        assert d.max() == 0, (tag, np.bincount(d.ravel())[:4])
    else:
        assert d.max() <= 1 and (d == 0).mean() >= 0.94, (tag, np.bincount(d.ravel())[:4])


def test_scenes_beyond_one_lsb_are_the_filters_weights():
    """The exported scenes held to `max ≤ 2` above (two chained bilinear fetches of 8-bit textures: BASELINE config 2's own path,
    multipass, the dynamics demo): max ≤ 1 under the llvmpipe filter model, and 96-98 % of the values identical (60-75 % without)"""
    street = synth.background_image(480, 270)
    P = np.load(Path(__file__).parent/"golden"/"pipeline.npz")
    fps, samplerate, frames = float(P["meta"][0]), int(P["meta"][1]), int(P["meta"][2])
    pcm, background = i16_to_f32(P["pcm_i16"]), synth.background_image(240, 135, seed=7)
    with O.llvmpipe_filter():
        cases = {"multipass": R.multipass_scene(street, 128, 72, 1, 60.0, 3),
                 "dynamics": R.dynamics_scene(street, 128, 72, 60.0, 90, (0, 1, 30, 59, 61, 89)),
                 "visualizer.ssaa1": R.audio_scene("visualizer", pcm, samplerate, background, 192, 108, 1, 2, fps, 60, pick=(1, 30, 59)),
                 "visualizer": R.audio_scene("visualizer", pcm, samplerate, background, 192, 108, 2, 2, fps, frames, pick=(0, 1, 10, 40, 99, frames - 1))}
    for tag, want in cases.items():
        d = np.abs(G[f"scene.{tag}.frames"].astype(int) - want.astype(int))
        assert d.max() <= 1 and (d == 0).mean() >= 0.95, (tag, np.bincount(d.ravel())[:4])


def test_anisotropic_extension_is_a_recorded_deviation():
    """texture.py:280 asks for 16x anisotropy on every texture. Where the context exposes EXT_texture_filter_anisotropic, llvmpipe
    answers with another filter for EVERY fetch of such a texture (also magnified, non-mipmapped ones); mesa.npz was rendered with the
    extension masked = the filter OpenGL 3.3 core specifies and hardware keeps for these isotropic footprints. filter.npz holds two
    probes with the extension ON and the measured distance — the decision as data: the oracle (and the kernels) follow the isotropic
    image and are far from the anisotropic one."""
    assert float(F["aniso.max"]) == 16.0
    differ, beyond, worst = F["aniso.visualizer.v0.5.image.deviation"]
    assert differ > 0.5 and beyond > 0.45 and worst > 200                     # 54 % of the Visualizer's values move, 49 % by more than 1 LSB
    assert F["aniso.default.plain.image.deviation"][0] == 0                   # an untextured fragment is untouched …
    assert F["aniso.default.plain.final.deviation"][1] > 0.05                 # … until final.glsl samples iScreen (anisotropy 16 there too)
    u, arrays, params = visualizer_inputs(160, 90, seed=21, volume=0.5, bg_size=(120, 68))
    screen = O.render("visualizer", u, oracle_textures(arrays, params), 160, 90, threads=8)
    near = np.abs(G["visualizer.v0.5.image"].astype(int) - screen.astype(int))
    far = np.abs(F["aniso.visualizer.v0.5.image"].astype(int) - screen.astype(int))
    assert near.max() <= 1 and (far > 1).mean() > 0.45


# ---- the benchmark's configuration: 3840x2160 at 2x SSAA ---------------------------------------------------------------------------

def c3_inputs(name: str):
    """The inputs tests/golden/make_golden_mesa_4k.py rendered (restated here: the generator itself needs /root/reference)"""
    K = np.load(Path(__file__).parent/"golden"/"mesa_4k.npz")
    seed, volume, _ = K[f"{name}.args"]
    w, h, ssaa = (int(v) for v in K["size"])
    u, arrays, params = visualizer_inputs(w, h, seed=int(seed), volume=float(volume), bg_size=(1920, 1080))
    if name == "bench":
        arrays["background"] = np.ascontiguousarray(np.flipud(synth.background_image(1920, 1080)))
    u.iSSAA = float(ssaa)
    return K, u, arrays, params, w, h, ssaa


def edge_aware_check(got_row: np.ndarray, want_row: np.ndarray, screen_rows: np.ndarray, where) -> np.ndarray:
    """`max ≤ 1 LSB`, except where ONE of the four supersamples of a pixel sits on the other side of an edge in one of the two
    renderings: that moves the resolved value by a quarter of the supersample spread around the pixel. Returns the |difference| histogram."""
    d = np.abs(got_row.astype(int) - want_row.astype(int))
    block = screen_rows[:, :, :3].astype(int).reshape(2, -1, 2, 3)                # (sample row, pixel, sample column, channel)
    high, low = block.max(axis=(0, 2)), block.min(axis=(0, 2))
    for side in (-1, 1):                                                          # the edge may run between this pixel and its neighbour
        high, low = np.maximum(high, np.roll(high, side, axis=0)), np.minimum(low, np.roll(low, side, axis=0))
    allowed = np.maximum(1, (high - low)//4 + 2)
    assert (d <= allowed).all(), (where, np.argwhere(d > allowed)[:4].tolist(), int(d.max()))
    return np.bincount(np.minimum(d.ravel(), 3), minlength=4)


def edge_aware_frame(got: np.ndarray, want: np.ndarray, screen: np.ndarray, where, ssaa: int = 2) -> np.ndarray:
    """edge_aware_check for a whole frame: `max ≤ 1 LSB`, except where ONE supersample under the pixel's resolve footprint sits on
    the other side of an edge in one of the two renderings — then a value may move by a quarter of the spread of the samples under
    the pixel and its four neighbours (the edge may run between two pixels, along either axis). `screen`: the oracle's iScreen of
    the frame, (h·ssaa, w·ssaa, ≥3). Returns the |difference| histogram (0, 1, 2, ≥3)."""
    h, w = got.shape[:2]
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    block = screen[:, :, :3].reshape(h, ssaa, w, ssaa, 3)
    high, low = block.max(axis=(1, 3)).astype(np.int16), block.min(axis=(1, 3)).astype(np.int16)
    if ssaa == 1:                                                                 # final.glsl's 3 x 3 tent: the eight texels around count as well
        for axis in (0, 1):
            high = np.maximum(high, np.maximum(np.roll(high, 1, axis), np.roll(high, -1, axis)))
            low = np.minimum(low, np.minimum(np.roll(low, 1, axis), np.roll(low, -1, axis)))
    top, bottom = high.copy(), low.copy()
    for axis in (0, 1):
        for side in (-1, 1):
            top, bottom = np.maximum(top, np.roll(high, side, axis)), np.minimum(bottom, np.roll(low, side, axis))
    allowed = np.maximum(1, (top - bottom)//4 + 2)
    assert (d <= allowed).all(), (where, np.argwhere(d > allowed)[:4].tolist(), int(d.max()))
    return np.bincount(np.minimum(d.ravel(), 3), minlength=4)


def one_supersample_explains(screen: np.ndarray, r: int, x: int, reference_rgb: np.ndarray) -> bool:
    """Does ONE of the four supersamples of output pixel (r, x) taking the colour of a supersample in its 6x6 neighbourhood — i.e. landing
    on the other side of an edge — turn the box mean into the reference's value (within the 1 LSB everything else meets)?
    `screen`: the oracle's supersampled rows (at least rows 2r-2 … 2r+3 present)"""
    block = screen[2*r:2*r + 2, 2*x:2*x + 2, :3].astype(float).reshape(4, 3)
    hood = screen[max(0, 2*r - 2):2*r + 4, max(0, 2*x - 2):2*x + 4, :3].astype(float).reshape(-1, 3)
    for i in range(4):
        trial = np.repeat(block[None], len(hood), axis=0)
        trial[:, i] = hood
        if (np.abs(np.rint(trial.mean(axis=1)) - reference_rgb.astype(float)) <= 1).all(axis=1).any():
            return True
    return False


@pytest.mark.parametrize("name", ["noise", "bench"])
def test_every_value_beyond_one_lsb_at_the_benchmark_size_is_accounted_for(name):
    """VERDICT round 3, item 2. tests/golden/mesa_4k_outliers.npz lists, by coordinates, every value of the reference's two 4K frames
    that the oracle misses by more than 1 LSB (30 + 76 of 2.77 M with the float-weight filter the kernels use). Here the SET is
    recomputed on every other stored row and must be exactly the stored one; each such pixel must be explained by ONE supersample
    sitting on the other side of an edge (a bar's outline, the waveform strip) — except the few that are llvmpipe's 8-bit filter
    weights, which vanish when the oracle filters as llvmpipe does (rows with outliers are rendered a second time under that switch)."""
    K, u, arrays, params, w, h, ssaa = c3_inputs(name)
    L = np.load(Path(__file__).parent/"golden"/"mesa_4k_outliers.npz")
    textures = oracle_textures(arrays, params)
    rows = [(n, int(r)) for n, r in enumerate(K[f"{name}.rows"])][::2]
    subset = {r for _, r in rows}

    def stored(mode: str) -> set:
        return {(int(r), int(x), int(c)) for r, x, c, _, _ in L[f"{name}.{mode}"] if int(r) in subset}

    def found(r: int, n: int, llvmpipe: bool):
        def go():
            screen = O.render("visualizer", u, textures, w*ssaa, h*ssaa, rows=(max(0, r*ssaa - 2), min(h*ssaa, (r + 1)*ssaa + 2)), threads=8)
            return screen, O.resolve(screen, w, h, 2, rows=(r, r + 1))[r]
        if llvmpipe:
            with O.llvmpipe_filter():
                screen, want = go()
        else:
            screen, want = go()
        got = K[f"{name}.final"][n]
        d = np.abs(got.astype(int) - want.astype(int))
        return screen, got, {(r, int(x), int(c)) for x, c in np.argwhere(d > 1)}

    spec, fixed, unexplained = set(), set(), []
    for n, r in rows:
        screen, got, here = found(r, n, False)
        spec |= here
        if here or any(q[0] == r for q in stored("llvmpipe")):
            screen, _, there = found(r, n, True)
            fixed |= there
            for x in {x for _, x, _ in there}:
                if not one_supersample_explains(screen, r, x, got[x]):
                    unexplained.append((r, x))
    assert spec == stored("spec"), (sorted(spec ^ stored("spec"))[:8])
    assert fixed == stored("llvmpipe"), (sorted(fixed ^ stored("llvmpipe"))[:8])
    assert not unexplained, unexplained                              # what the filter model does not remove is one supersample across an edge
    assert len(spec - fixed) <= 3                                    # … and what it does remove is 2 LSB, the filter's weights
    assert all(d == 2 for r, x, c, d, _ in L[f"{name}.spec"] if (int(r), int(x), int(c)) in spec - fixed)


@pytest.mark.parametrize("name", ["noise", "bench"])
def test_benchmark_size_frames_rendered_by_the_reference(name):
    """Whole 3840x2160 2xSSAA frames exported by the reference on llvmpipe (every 13th / 27th row kept, all columns): ≥ 99.99 % of
    the values within 1 LSB of the oracle, none further than one supersample crossing a bar's edge explains (≈ 1 pixel in 10⁵)"""
    K, u, arrays, params, w, h, ssaa = c3_inputs(name)
    textures = oracle_textures(arrays, params)
    histogram = np.zeros(4, int)
    for n, r in list(enumerate(K[f"{name}.rows"]))[::2]:                           # every other stored row keeps the CPU suite short
        screen = O.render("visualizer", u, textures, w*ssaa, h*ssaa, rows=(r*ssaa, (r + 1)*ssaa), threads=8)
        want = O.resolve(screen, w, h, 2, rows=(r, r + 1))[r]
        histogram += edge_aware_check(K[f"{name}.final"][n], want, screen[r*ssaa:(r + 1)*ssaa], (name, int(r)))
    assert histogram[:2].sum()/histogram.sum() >= 0.9999, histogram
    # the fragment pass alone (every 16th supersample column of three bands): before final.glsl nothing sits on an averaged edge
    for first, last in K["bands"]:
        screen = O.render("visualizer", u, textures, w*ssaa, h*ssaa, rows=(first*ssaa, last*ssaa), threads=8)
        d = np.abs(screen[first*ssaa:last*ssaa, ::16].astype(int) - K[f"{name}.band{first}.screen"].astype(int))
        assert (d <= 1).mean() >= 0.9995, (name, int(first), np.bincount(d.ravel())[:6])
