"""
The pixel oracle against the REFERENCE'S GLSL executed by an independent OpenGL implementation (Google SwiftShader,
OpenGL ES 3.0): tests/golden/gles.npz holds what tests/golden/make_golden_gles.py rendered in the build container from
the shader files under /root/reference (assembled as shader.py:190-235 does, adapted mechanically to GLSL ES).
This is what pins the oracle's reading of the GLSL, the GL sampler and the varyings; CPU only.

Tolerance: 1 LSB per channel. Built-in precision (sin, pow, atan) and the sub-texel precision of the bilinear filter
are implementation choices in OpenGL, so two conforming implementations differ in the last bit of an 8-bit channel;
the north star's parity bar is that same 1 LSB. One image has a known ill-conditioned term (default.glsl's ring:
1/circle² next to circle = 0) and is held to 1 LSB on 99.8 % of its values.
"""
from pathlib import Path

import numpy as np
import pytest

from oracle import binding as O
from tests.helpers import oracle_textures, visualizer_inputs

G = np.load(Path(__file__).parent/"golden"/"gles.npz")
CAMERAS = {"plain": {}, "moved": dict(iCameraZoom=1.3, iCameraIsometric=0.2, iCameraPosition=(0.1, -0.05, 0.0)),
           "stereo": dict(iCameraProjection=1, iCameraSeparation=0.07, iCameraZoom=1.2), "equirect": dict(iCameraProjection=2, iCameraZoom=0.8)}


def agree(tag: str, want: np.ndarray, fraction: float = 1.0) -> None:
    got = G[f"{tag}.image"]
    assert got.shape == want.shape, (tag, got.shape, want.shape)
    d = np.abs(got.astype(int) - want.astype(int))
    within = (d <= 1).mean()
    assert within >= fraction, f"{tag}: {100*within:.3f}% within 1 LSB (max {d.max()})"
    assert np.abs(got.astype(float).mean() - want.astype(float).mean()) < 0.25, tag       # no systematic offset


@pytest.mark.parametrize("camera", list(CAMERAS))
def test_default_fragment_every_projection(camera):
    u = O.default_uniforms(160, 90, iTime=0.75, iTau=0.3, **CAMERAS[camera])
    agree(f"default.{camera}", O.render("default", u, {}, 160, 90, threads=4), fraction=0.998 if camera == "plain" else 1.0)


def test_untextured_fragments():
    u = O.default_uniforms(96, 54, iTime=3.0, iTau=0.3)
    agree("missing", O.render("missing", u, {}, 96, 54))
    agree("shadertoy", O.render("shadertoy", u, {}, 96, 54))
    agree("multi_child", O.render("multi_child", O.default_uniforms(64, 36), {}, 64, 36))
    agree("raymarch", O.render("raymarch", O.default_uniforms(160, 90), {}, 160, 90, threads=4))
    agree("raymarch.moved", O.render("raymarch", O.default_uniforms(160, 90, iCameraPosition=(0.4, 0.2, -1.5), iCameraZoom=0.8), {}, 160, 90, threads=4))
    agree("mandelbrot", O.render("mandelbrot", O.default_uniforms(160, 90, iQuality=0.2), {}, 160, 90, threads=4))


@pytest.mark.parametrize("volume", [0.0, 0.5, 1.2])
def test_visualizer_with_its_radial_blur(volume):
    """The benchmark fragment: 91 bilinear taps with float loop counters, rotated bars, nearest spectrogram, linear waveform"""
    u, arrays, params = visualizer_inputs(160, 90, seed=21, volume=volume, bg_size=(120, 68))
    agree(f"visualizer.v{volume}", O.render("visualizer", u, oracle_textures(arrays, params), 160, 90, threads=8))


def test_audio_texture_fragments():
    u, arrays, params = visualizer_inputs(128, 72, seed=5)
    arrays["iSpectrogram"] = arrays["iSpectrogram"]*3
    for name in ("bars", "waveform"):
        agree(name, O.render(name, u, oracle_textures(arrays, params), 128, 72, threads=4))
    u.user[0] = 0.35
    agree("dynamics", O.render("dynamics", u, oracle_textures(arrays, params), 128, 72))


@pytest.mark.parametrize("filter", ["nearest", "linear"])
@pytest.mark.parametrize("wrap", ["clamp", "repeat"])
def test_sampler_addressing_and_filtering(filter, wrap):
    """texture() on a 7x5 RGBA8 grid at coordinates from -0.75 to 1.75: texel addressing, wrap modes, bilinear weights"""
    texture = O.make_texture(G["sampler.texels"], filter, wrap == "repeat", wrap == "repeat")
    w, h = 70, 50
    want = np.zeros((h, w, 4), np.uint8)
    for j in range(h):
        for i in range(w):
            # astuv of the pixel centre (vertex/default.glsl:9-10), then the probe's `astuv*2.5 - 0.75`
            s = np.float32(((np.float32(i) + np.float32(0.5))/np.float32(w)*np.float32(2) - np.float32(1) + np.float32(1))/np.float32(2))
            t = np.float32(((np.float32(j) + np.float32(0.5))/np.float32(h)*np.float32(2) - np.float32(1) + np.float32(1))/np.float32(2))
            c = O.sample(texture, s*np.float32(2.5) - np.float32(0.75), t*np.float32(2.5) - np.float32(0.75))
            want[j, i] = np.rint(np.clip(c, 0, 1)*255)
    agree(f"sampler.{filter}.{wrap}", want)
    if filter == "nearest":
        assert np.array_equal(G[f"sampler.{filter}.{wrap}.image"], want)                  # no filtering arithmetic: identical


def test_layers_history_and_final():
    w, h = 128, 72
    background = G["multipass.background"]
    u = O.default_uniforms(w, h, iLayer=0)
    layer0 = O.render("multipass", u, {"background": O.make_texture(background)}, w, h, threads=4)
    agree("multipass.layer0", layer0)
    u.iLayer = 1
    # layer 1 samples what GL rendered into iScreen0x0, so feed the oracle the same texels
    first = G["multipass.layer0.image"]
    agree("multipass.layer1", O.render("multipass", u, {"background": O.make_texture(background), 0: O.make_texture(first, "linear", False, False)}, w, h, threads=4))
    history = G["motionblur.history"]
    u = O.default_uniforms(96, 54, iLayer=1)
    u.user[0] = len(history)
    agree("motionblur.layer1", O.render("motionblur", u, {t: O.make_texture(history[t], "linear", False, False) for t in range(len(history))}, 96, 54, threads=4))
    screen = G["final.screen"]
    for (fw, fh, sub) in ((64, 36, 2), (64, 36, 1), (128, 72, 2), (32, 18, 4)):
        want = O.resolve(screen, fw, fh, sub)
        got = G[f"final.{fw}x{fh}.k{sub}.image"]
        d = np.abs(got[..., :3].astype(int) - want.astype(int))
        assert d.max() <= 1, (fw, fh, sub, d.max())
        assert (got[..., 3] == 255).all()


@pytest.mark.parametrize("tag,kw", [("tetration", {}), ("tetration.zoomed", dict(iCameraZoom=2.5, iCameraPosition=(-0.7, 0.1, 0.0)))])
def test_tetration_integer_division(tag, kw):
    """`it / MAX_STEPS` is an integer division (value 0 or 1). The iteration is chaotic along the fractal's boundary, where
    the built-ins' last bits decide between escape and hue: 99.5 % of the values agree to 1 LSB, the rest lies on that boundary."""
    agree(tag, O.render("tetration", O.default_uniforms(160, 90, **kw), {}, 160, 90, threads=4), fraction=0.995)


def test_life_simulation_and_visuals():
    """life/simulation.glsl compiled WITH its integer types (texelFetch, int arrays, %): the hold branch is identical; the rule
    branch is identical away from the border. On the border ring the two differ by design: texelFetch outside the texture is
    undefined in OpenGL — SwiftShader clamps the coordinate, the oracle (and the HIP kernel) read zero like robust-access
    desktop drivers do (sfo_pixel.c texel_fetch)."""
    states = G["life.states"]
    lh, lw = states[1].shape[:2]
    for frame in (0, 6, 7):
        u = O.default_uniforms(lw, lh, iFrame=frame)
        u.user[0], u.user[1], u.user[2] = lw, lh, 6
        want = O.render_to("life_simulation", u, {1: O.make_texture(states[1], "nearest", True, True)}, lw, lh, 1, np.float32)[..., 0] > 0.5
        got = G[f"life_simulation.f{frame}.image"][..., 0] > 127
        if frame % 6:
            assert np.array_equal(got, want)
        else:
            assert np.array_equal(got[1:-1, 1:-1], want[1:-1, 1:-1])
            clamped = np.pad(states[1][..., 0], 1, mode="edge")
            near = sum(clamped[1 + dy:1 + dy + lh, 1 + dx:1 + dx + lw] for dx in (-1, 0, 1) for dy in (-1, 0, 1) if (dx, dy) != (0, 0))
            assert np.array_equal(got, np.where(states[1][..., 0] == 1, (near == 2) | (near == 3), near == 3))     # SwiftShader = clamp-to-edge fetch
    visuals = O.render("life_visuals", O.default_uniforms(128, 72, iCameraZoom=0.9),
                       {t: O.make_texture(states[t], "nearest", True, True) for t in range(5)}, 128, 72, threads=4)
    assert np.array_equal(G["life_visuals.image"], visuals)


def test_end_to_end_frames_from_the_reference_audio_state():
    """The whole per-frame path of the reference on real inputs: its numpy audio state for a 1.5 s clip (pipeline.npz, captured
    from its own code) drives its own GLSL — visualizer.frag at 2x SSAA, then final.glsl — on SwiftShader; the oracle, fed the
    same audio state, produces the same RGB8 frames within 1 LSB."""
    from shaderflow_amd import synth
    P = np.load(Path(__file__).parent/"golden"/"pipeline.npz")
    fps, frames = float(P["meta"][0]), int(P["meta"][2])
    w, h, ssaa = (int(v) for v in G["frames.size"])
    runtime, bins = frames/fps, int(P["bins"][0])
    background = O.make_texture(np.flipud(synth.background_image(240, 135, seed=7)))
    for k in G["frames.index"]:
        t = float(P["time"][k])
        u = O.default_uniforms(w, h, iTime=t, iTau=(t/runtime) % 1.0, iDuration=runtime, iSSAA=float(ssaa), iFramerate=fps, iFrame=round(t*fps),
                               iAudioVolume=float(P["vol_value"][k]), iAudioVolumeIntegral=float(P["vol_integral"][k]), iAudioSTD=float(P["std_value"][k]),
                               iSpectrogramLength=1, iSpectrogramBins=bins, iWaveformLength=180)
        textures = {"background": background,
                    "iSpectrogram": O.make_texture(np.ascontiguousarray(P["spec_value"][k]).reshape(bins, 1, 2), "nearest", True, False),
                    "iWaveform": O.make_texture(np.ascontiguousarray(P["wave_row"][k]).reshape(1, 180, 2), "linear", False, False)}
        want = O.resolve(O.render("visualizer", u, textures, w*ssaa, h*ssaa, threads=8), w, h, 2, threads=4)
        got = G[f"frames.{k}"]
        d = np.abs(got.astype(int) - want.astype(int))
        assert d.max() <= 1, (int(k), d.max(), (d > 1).sum())


def test_benchmark_size_bands_against_the_reference_glsl():
    """gles_4k.npz: three bands of the 3840x2160 2xSSAA frame rendered from the reference's GLSL (make_golden_gles_4k.py). The
    fragment pass (every 16th supersample column is stored) within 1 LSB everywhere; after final.glsl >= 99.8 % of the values
    within 1 LSB, the rest on antialiased outlines where one supersample of four lands on the other side of a bar's edge."""
    from pathlib import Path
    from tests.helpers import oracle_textures, visualizer_inputs
    K = np.load(Path(__file__).parent/"golden"/"gles_4k.npz")
    w, h, ssaa, seed, volume = int(K["args"][0]), int(K["args"][1]), int(K["args"][2]), int(K["args"][3]), float(K["args"][4])
    u, arrays, params = visualizer_inputs(w, h, seed=seed, volume=volume, bg_size=(int(K["args"][5]), int(K["args"][6])))
    u.iSSAA = float(ssaa)
    for first, last in K["bands"]:
        screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, rows=(first*ssaa, last*ssaa), threads=8)
        d = np.abs(screen[first*ssaa:last*ssaa, ::16].astype(int) - K[f"rows{first}.screen"].astype(int))
        assert (d <= 1).mean() >= 0.9995, (int(first), np.bincount(d.ravel())[:6])
        final = O.resolve(screen, w, h, 2, rows=(first, last), threads=8)[first:last]
        d = np.abs(final.astype(int) - K[f"rows{first}.final"].astype(int))
        assert (d <= 1).mean() >= 0.998, (int(first), np.bincount(d.ravel())[:6])
