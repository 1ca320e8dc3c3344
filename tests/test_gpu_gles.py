"""
The HIP kernels directly against the reference's GLSL as rendered by an independent OpenGL implementation (tests/golden/gles.npz,
see tests/test_oracle_gles.py): no oracle in between. Every kernel — the LDS-tiled visualizer kernels included — is held to 1 LSB
per channel, what two GL implementations agree to (measured: profiles/r02_parity_histogram.txt has no value further off at these
sizes). At the benchmark's own size (gles_4k.npz: bands of a 3840x2160 2xSSAA frame) 0.2 % of the values sit on antialiased
outlines where the two implementations put a supersample on different sides; the test pins those to the parity oracle.
"""
from pathlib import Path

import numpy as np
import pytest

from oracle import binding as O
from shaderflow_amd import _native as N
from tests.helpers import Gpu, gpu_bind_all, visualizer_inputs
from tests.test_oracle_gles import CAMERAS

pytestmark = pytest.mark.gpu
G = np.load(Path(__file__).parent/"golden"/"gles.npz")


@pytest.fixture()
def gpu():
    g = Gpu()
    yield g
    g.close()


def close_to(tag: str, got: np.ndarray, bound: int = 1, fraction: float = 1.0) -> None:
    want = G[f"{tag}.image"]
    d = np.abs(got.astype(int) - want[..., :got.shape[2]].astype(int))
    assert (d <= bound).mean() >= fraction, f"{tag}: max {d.max()}, {100*(d <= bound).mean():.3f}% within {bound}"


@pytest.mark.parametrize("camera", list(CAMERAS))
def test_default_fragment(gpu, camera):
    u = O.default_uniforms(160, 90, iTime=0.75, iTau=0.3, **CAMERAS[camera])
    prog, _ = gpu.program("default")
    gpu.set_uniforms(prog, u)
    close_to(f"default.{camera}", gpu.render(prog, 160, 90), fraction=0.998 if camera == "plain" else 1.0)


@pytest.mark.parametrize("volume", [0.0, 0.5, 1.2])
def test_visualizer_tiled_kernel(gpu, volume):
    u, arrays, params = visualizer_inputs(160, 90, seed=21, volume=volume, bg_size=(120, 68))
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    got = gpu.render(prog, 160, 90)
    want = G[f"visualizer.v{volume}.image"]
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1, (d.max(), np.bincount(d.ravel()))                      # the north star's bound, against the reference's GLSL


def test_benchmark_kernel_against_the_reference_glsl_at_4k(gpu):
    """The kernel bench.py times (k_visualizer_fast: per-frame column/row tables, axis lines, 72x10-cell LDS tile) is only selected
    at sizes like the benchmark's, so it gets goldens of its own: three bands of four rows of the 3840x2160 2xSSAA frame of
    test_full_size_properties_4k_ssaa2, rendered from the reference's visualizer.frag + final.glsl by SwiftShader
    (tests/golden/make_golden_gles_4k.py). >= 99.8 % of the values within 1 LSB; the rest are antialiased outlines of the bars
    where the two GL implementations put one of the four supersamples on different sides — there the kernel must agree with the
    parity oracle (the bit-exact chain), which differs from the GLSL rendering at the same places."""
    from tests.helpers import oracle_textures
    K = np.load(Path(__file__).parent/"golden"/"gles_4k.npz")
    w, h, ssaa, seed, volume = int(K["args"][0]), int(K["args"][1]), int(K["args"][2]), int(K["args"][3]), float(K["args"][4])
    u, arrays, params = visualizer_inputs(w, h, seed=seed, volume=volume, bg_size=(int(K["args"][5]), int(K["args"][6])))
    u.iSSAA = float(ssaa)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    frame = gpu.render_resolve(prog, w, h, ssaa, 2)
    assert gpu.lib.sfx_last_kernel().decode().startswith("k_visualizer_"), gpu.lib.sfx_last_kernel()
    for first, last in K["bands"]:
        want = K[f"rows{first}.final"]
        got = frame[first:last]
        d = np.abs(got.astype(int) - want.astype(int))
        assert (d <= 1).mean() >= 0.998, (int(first), np.bincount(d.ravel())[:6])
        screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, rows=(first*ssaa, last*ssaa), threads=8)
        oracle = O.resolve(screen, w, h, 2, rows=(first, last), threads=8)[first:last]
        assert np.abs(got.astype(int) - oracle.astype(int)).max() <= 1
        far = d >= 2
        assert (np.abs(oracle.astype(int) - want.astype(int))[far] >= 1).all()  # where the kernel is off the GLSL, so is the bit-exact chain


def test_audio_fragments_and_raymarch(gpu):
    u, arrays, params = visualizer_inputs(128, 72, seed=5)
    arrays["iSpectrogram"] = arrays["iSpectrogram"]*3
    for name in ("bars", "waveform"):
        prog, _ = gpu.program(name)
        gpu.set_uniforms(prog, u)
        gpu_bind_all(gpu, prog, {k: v for k, v in arrays.items() if k != "background"}, params)
        close_to(name, gpu.render(prog, 128, 72))
    for tag, kw in (("raymarch", {}), ("raymarch.moved", dict(iCameraPosition=(0.4, 0.2, -1.5), iCameraZoom=0.8))):
        prog, _ = gpu.program("raymarch")
        gpu.set_uniforms(prog, O.default_uniforms(160, 90, **kw))
        close_to(tag, gpu.render(prog, 160, 90))
    prog, _ = gpu.program("mandelbrot")
    gpu.set_uniforms(prog, O.default_uniforms(160, 90, iQuality=0.2))
    close_to("mandelbrot", gpu.render(prog, 160, 90))


def test_layers_history_and_final(gpu):
    w, h = 128, 72
    background = G["multipass.background"]
    prog, _ = gpu.program("multipass")
    gpu.set_uniforms(prog, O.default_uniforms(w, h))
    assert gpu.bind(prog, "background", gpu.texture(background))
    close_to("multipass.layer0", gpu.render(prog, w, h, layer=0))
    assert gpu.bind(prog, "iScreen0x0", gpu.texture(G["multipass.layer0.image"], "linear", False, False))
    close_to("multipass.layer1", gpu.render(prog, w, h, layer=1))
    history = G["motionblur.history"]
    prog, _ = gpu.program("motionblur")
    gpu.set_uniforms(prog, O.default_uniforms(96, 54))
    assert gpu.set_values(prog, "iScreenTemporal", len(history), integer=True)
    for t in range(len(history)):
        assert gpu.bind(prog, f"iScreen{t}x0", gpu.texture(history[t], "linear", False, False))
    close_to("motionblur.layer1", gpu.render(prog, 96, 54, layer=1))
    for (fw, fh, sub) in ((64, 36, 2), (64, 36, 1), (128, 72, 2), (32, 18, 4)):
        close_to(f"final.{fw}x{fh}.k{sub}", gpu.resolve(G["final.screen"], fw, fh, sub))


def test_life(gpu):
    states = G["life.states"]
    lh, lw = states[1].shape[:2]
    prog, _ = gpu.program("life_simulation")
    for frame in (0, 6, 7):
        u = O.default_uniforms(lw, lh, iFrame=frame)
        gpu.set_uniforms(prog, u)
        assert gpu.set_values(prog, "iLifeSize", (lw, lh)) and gpu.set_values(prog, "iLifePeriod", 6, integer=True)
        assert gpu.bind(prog, "iLife1x0", gpu.texture(states[1], "nearest", True, True))
        got = gpu.render(prog, lw, lh, comps=1, dtype=np.float32)[..., 0] > 0.5
        want = G[f"life_simulation.f{frame}.image"][..., 0] > 127
        inner = (slice(1, -1), slice(1, -1)) if frame % 6 == 0 else (slice(None), slice(None))     # texelFetch outside: see test_oracle_gles
        assert np.array_equal(got[inner], want[inner]), frame
    prog, _ = gpu.program("life_visuals")
    gpu.set_uniforms(prog, O.default_uniforms(128, 72, iCameraZoom=0.9))
    for t in range(5):
        assert gpu.bind(prog, f"iLife{t}x0", gpu.texture(states[t], "nearest", True, True))
    assert np.array_equal(gpu.render(prog, 128, 72), G["life_visuals.image"])


@pytest.mark.parametrize("batch", [None, False])
def test_end_to_end_export_against_the_reference_pipeline(batch):
    """The north star's parity statement, end to end and with no oracle in between: the product exports the Visualizer scene from
    PCM (STFT, filterbank, DynamicNumbers, waveform and loudness on the device; fused fragment + resolve), and the frames are
    compared with what the REFERENCE's own numpy audio code (pipeline.npz) fed through the REFERENCE's own GLSL (SwiftShader:
    visualizer.frag at 2x SSAA, then final.glsl) produced. Bound: 1 LSB per channel on every value."""
    from examples.scenes import Visualizer, make
    from shaderflow_amd import synth
    from tests.helpers import i16_to_f32
    P = np.load(Path(__file__).parent/"golden"/"pipeline.npz")
    fps, samplerate, frames = float(P["meta"][0]), int(P["meta"][1]), int(P["meta"][2])
    w, h, ssaa = (int(v) for v in G["frames.size"])
    scene = make(Visualizer, audio=(i16_to_f32(P["pcm_i16"]), samplerate), background=synth.background_image(240, 135, seed=7))
    raw = scene.main(width=w, height=h, fps=fps, ssaa=ssaa, subsample=2, time=frames/fps, output=bytes, batch=batch)
    got = np.frombuffer(raw, np.uint8).reshape(-1, h, w, 3)
    assert got.shape[0] == frames
    for k in G["frames.index"]:
        want = G[f"frames.{k}"]
        d = np.abs(got[k].astype(int) - want.astype(int))
        assert d.max() <= 1, (int(k), d.max(), np.bincount(d.ravel()))
