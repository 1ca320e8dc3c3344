"""
The HIP path against THE REFERENCE ITSELF, no oracle in between: tests/golden/mesa.npz / mesa_4k.npz hold what /root/reference's own
Python rendered through Mesa llvmpipe (OpenGL 4.5 core) with its GLSL unmodified — see tests/test_oracle_mesa.py and
tests/golden/make_golden_mesa.py. Kernels alone on the probes' inputs, the product's example scenes (examples/scenes.py: the
reference's demo.py classes on the product's API) exported end to end, and the benchmark's configuration whole-frame.

Bounds are those of tests/test_oracle_mesa.py: `max ≤ 1` where the reference's own rendering allows it; `max ≤ 2` with ≥ 98.5 %
within 1 where an 8-bit texture goes through llvmpipe's 8-fractional-bit bilinear filter twice; at 4K the edge-aware bound (a value
may differ by what ONE supersample crossing a bar's edge explains).
"""
from pathlib import Path

import numpy as np
import pytest

from oracle import binding as O
from shaderflow_amd import synth
from tests.helpers import Gpu, gpu_bind_all, i16_to_f32, oracle_textures, visualizer_inputs
from tests.test_oracle_mesa import CAMERAS, c3_inputs, edge_aware_check

pytestmark = pytest.mark.gpu
G = np.load(Path(__file__).parent/"golden"/"mesa.npz")


@pytest.fixture()
def gpu():
    g = Gpu()
    yield g
    g.close()


def close_to(tag: str, got: np.ndarray, bound: int = 1, fraction: float = 1.0, key: str = "image") -> None:
    want = G[f"{tag}.{key}"]
    d = np.abs(got.astype(int) - want[..., :got.shape[2]].astype(int))
    assert d.max() <= bound and (d <= 1).mean() >= fraction, f"{tag}: max {d.max()}, {100*(d <= 1).mean():.3f}% within 1"


# ---- kernels on the probes' inputs ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("camera", list(CAMERAS))
def test_default_fragment(gpu, camera):
    u = O.default_uniforms(160, 90, iTime=0.75, iTau=0.3, **CAMERAS[camera])
    prog, _ = gpu.program("default")
    gpu.set_uniforms(prog, u)
    bound, fraction = {"plain": (8, 0.996), "stereo": (8, 0.999)}.get(camera, (1, 1.0))        # the 1/circle² ring
    close_to(f"default.{camera}", gpu.render(prog, 160, 90), bound, fraction)


@pytest.mark.parametrize("volume", [0.0, 0.5, 1.2])
def test_visualizer_kernels(gpu, volume):
    u, arrays, params = visualizer_inputs(160, 90, seed=21, volume=volume, bg_size=(120, 68))
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    close_to(f"visualizer.v{volume}", gpu.render(prog, 160, 90), *((2, 0.999) if volume == 0.0 else (1, 1.0)))


def test_visualizer_fused_with_the_resolve(gpu):
    u, arrays, params = visualizer_inputs(192, 108, seed=33, volume=0.9, bg_size=(160, 90))
    u.iSSAA = 2.0
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    close_to("visualizer.ssaa2", gpu.render(prog, 384, 216))
    close_to("visualizer.ssaa2", gpu.render_resolve(prog, 192, 108, 2, 2), key="final")


def test_other_fragments(gpu):
    u, arrays, params = visualizer_inputs(128, 72, seed=5)
    arrays["iSpectrogram"] = arrays["iSpectrogram"]*3
    for name in ("bars", "waveform"):
        prog, _ = gpu.program(name)
        gpu.set_uniforms(prog, u)
        gpu_bind_all(gpu, prog, {k: v for k, v in arrays.items() if k != "background"}, params)
        assert np.array_equal(gpu.render(prog, 128, 72), G[f"{name}.image"]), name
    for tag, name, kw in (("raymarch", "raymarch", {}), ("raymarch.moved", "raymarch", dict(iCameraPosition=(0.4, 0.2, -1.5), iCameraZoom=0.8)),
                          ("shadertoy", "shadertoy", None), ("multi_child", "multi_child", None)):
        prog, _ = gpu.program(name)
        size = {"shadertoy": (96, 54), "multi_child": (64, 36)}.get(tag, (160, 90))
        gpu.set_uniforms(prog, O.default_uniforms(*size, iTime=3.0, iTau=0.3) if tag == "shadertoy" else O.default_uniforms(*size, **(kw or {})))
        assert np.array_equal(gpu.render(prog, *size), G[f"{tag}.image"]), tag
    prog, _ = gpu.program("mandelbrot")
    gpu.set_uniforms(prog, O.default_uniforms(160, 90, iQuality=0.2))
    close_to("mandelbrot", gpu.render(prog, 160, 90), 2, 0.9999)
    for (fw, fh, sub) in ((64, 36, 2), (64, 36, 1), (128, 72, 2), (32, 18, 4)):
        close_to(f"final.{fw}x{fh}.k{sub}", gpu.resolve(G["final.screen"], fw, fh, sub), 1 if fw*sub == 128 or sub == 1 else 2, 0.99)


# ---- the example scenes, exported end to end ---------------------------------------------------------------------------------------

def export(scene, tag: str, **kw) -> np.ndarray:
    width, height, ssaa, subsample, fps, frames = G[f"scene.{tag}.args"]
    raw = scene.main(width=int(width), height=int(height), ssaa=(int(ssaa) if ssaa == int(ssaa) else float(ssaa)), subsample=int(subsample),
                     fps=float(fps), time=int(frames)/float(fps), output=bytes, **kw)
    got = np.frombuffer(raw, np.uint8).reshape(-1, int(height), int(width), 3)
    assert got.shape[0] == int(frames)
    return got[G[f"scene.{tag}.index"]]


def same_frames(tag: str, got: np.ndarray, bound: int = 1, fraction: float = 1.0) -> None:
    want = G[f"scene.{tag}.frames"]
    for n, k in enumerate(G[f"scene.{tag}.index"]):
        d = np.abs(got[n].astype(int) - want[n].astype(int))
        assert d.max() <= bound and (d <= 1).mean() >= fraction, f"scene.{tag} frame {k}: max {d.max()}, {100*(d <= 1).mean():.3f}% within 1"


@pytest.mark.parametrize("batch", [None, False])
def test_clock_scenes(batch):
    import examples.scenes as S
    same_frames("basic", export(S.Basic(), "basic", batch=batch))                     # BASELINE config 1
    same_frames("shadertoy", export(S.ShaderToy(), "shadertoy", batch=batch))
    same_frames("raymarch", export(S.RayMarch(), "raymarch", batch=batch))
    same_frames("multishader", export(S.MultiShader(), "multishader", batch=batch))


def test_layered_temporal_and_scripted_scenes():
    import examples.scenes as S
    same_frames("multipass", export(S.Multipass(), "multipass"), 2, 0.985)
    same_frames("motionblur", export(S.MotionBlur(), "motionblur"))
    same_frames("dynamics", export(S.Dynamics(), "dynamics"), 2, 0.99)
    np.random.seed(int(G["scene.life.seed"][0]))
    same_frames("life", export(S.Life(), "life"))


def test_scene_that_drives_the_camera():
    """The same scene class on both sides (move / zoom / rotate2d / projection switch from update()): the product's camera systems
    emit the reference's uniform values frame by frame (float64 DynamicNumbers on the host, as in the reference), and its frames
    are the reference's within the default fragment's bound"""
    from examples.scenes import Basic
    from shaderflow_amd.camera import CameraProjection
    from shaderflow_amd.module import ShaderModule
    snapshots = []

    class Snapshot(ShaderModule):
        def update(self):
            snapshots.append({v.name: np.array(v.value, dtype=np.float64).ravel() for v in self.scene.shader.full_pipeline()
                              if v.type != "sampler2D" and v.value is not None})

    class Moving(Basic):
        frame_count = 0

        def build(self):
            Snapshot(scene=self)

        def update(self):
            self.camera.move(np.array([0.02, -0.01, 0.0]))
            self.camera.apply_zoom(0.05)
            self.camera.rotate2d(3.0)
            if self.frame_count == 3:
                self.camera.projection = CameraProjection.Stereoscopic
            self.frame_count += 1

    got = export(Moving(), "moving_camera")
    want = G["scene.moving_camera.frames"]
    for k in range(6):
        d = np.abs(got[k].astype(int) - want[k].astype(int))
        assert (d <= 1).mean() >= 0.99, (k, d.max())                        # default.glsl's ring (1/circle^2 next to zero)
    names, sizes, values = G["scene.moving_camera.uniform_names"], G["scene.moving_camera.uniform_sizes"], G["scene.moving_camera.uniforms"]
    for k in range(6):
        at = 0
        for name, size in zip(names, sizes):
            if str(name).startswith("iCamera"):
                mine = snapshots[k][str(name)]
                assert np.allclose(mine, values[k][at:at + size], rtol=1e-12, atol=1e-14), (k, str(name), mine, values[k][at:at + size])
            at += size


@pytest.mark.parametrize("batch", [None, False])
def test_audio_scenes_from_pcm_to_frames(batch):
    """The north star's parity statement with the reference on the other side: the product exports the scene from PCM (STFT,
    filterbank, DynamicNumbers, waveform, loudness on the device; fused fragment + resolve; frame tape and frame loop), the
    reference did the same with numpy + GLSL on llvmpipe"""
    import examples.scenes as S
    P = np.load(Path(__file__).parent/"golden"/"pipeline.npz")
    audio = (i16_to_f32(P["pcm_i16"]), int(P["meta"][1]))
    background = synth.background_image(240, 135, seed=7)
    same_frames("visualizer", export(S.make(S.Visualizer, audio=audio, background=background), "visualizer", batch=batch), 2, 0.9995)
    same_frames("visualizer.ssaa1", export(S.make(S.Visualizer, audio=audio, background=background), "visualizer.ssaa1", batch=batch), 2, 0.99)
    same_frames("musicbars", export(S.make(S.MusicBars, audio=audio), "musicbars", batch=batch))
    same_frames("waveform", export(S.make(S.Waveform, audio=audio), "waveform", batch=batch))


@pytest.mark.parametrize("smooth", [False, True])
@pytest.mark.parametrize("batch", [None, False])
def test_scrolling_spectrogram_through_the_translator(smooth, batch):
    """ShaderSpectrogram(length = 0.5 s): a 30-column texture, one column rewritten per frame, shown by a fragment of this repository's
    own. The reference compiled the text with its GL driver and scrolled its numpy spectrogram; here the text goes through glsl2hip and
    the frame tape keeps one texture state per frame of a batch (k_spectrogram_scroll). Same frames, tape and frame loop."""
    from shaderflow_amd import ShaderScene
    from shaderflow_amd.audio import ShaderAudio
    from shaderflow_amd.audio.spectrogram import ShaderSpectrogram
    from shaderflow_amd.piano import PianoNote
    from tests.helpers import SCROLL_FRAGMENT
    pcm = synth.sweep_clip(2.0, 44100)

    class Scroller(ShaderScene):
        def build(self):
            super().build()
            self.audio = ShaderAudio(scene=self, name="iAudio")
            self.audio.load(samples=pcm, samplerate=44100)
            self.spectrogram = ShaderSpectrogram(scene=self, audio=self.audio, length=0.5, smooth=smooth)
            self.spectrogram.from_notes(start=PianoNote.from_frequency(20), end=PianoNote.from_frequency(14000), piano=True)
            self.shader.fragment = SCROLL_FRAGMENT

    tag = f"scroller.{'smooth' if smooth else 'nearest'}"
    same_frames(tag, export(Scroller(), tag, batch=batch))


# ---- the benchmark's configuration --------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["noise", "bench"])
def test_benchmark_kernel_against_whole_frames_of_the_reference(gpu, name):
    """k_visualizer_strip at 3840x2160 2xSSAA against the frame the reference exported on llvmpipe: all 3 840 columns of every
    13th / 27th row (every row phase of a block, every block column) and three full bands"""
    K, u, arrays, params, w, h, ssaa = c3_inputs(name)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    frame = gpu.render_resolve(prog, w, h, ssaa, 2)
    assert gpu.lib.sfx_last_kernel().decode().startswith("k_visualizer_strip<"), gpu.lib.sfx_last_kernel()
    textures = oracle_textures(arrays, params)
    histogram = np.zeros(4, int)
    for n, r in enumerate(K[f"{name}.rows"]):
        screen = O.render("visualizer", u, textures, w*ssaa, h*ssaa, rows=(r*ssaa, (r + 1)*ssaa), threads=16)      # only for the edge mask
        histogram += edge_aware_check(frame[r], K[f"{name}.final"][n], screen[r*ssaa:(r + 1)*ssaa], (name, int(r)))
    assert histogram[:2].sum()/histogram.sum() >= 0.9999, histogram
    for first, last in K["bands"]:
        d = np.abs(frame[first:last].astype(int) - K[f"{name}.band{first}.final"].astype(int))
        assert (d <= 1).mean() >= 0.9995, (name, int(first), np.bincount(d.ravel())[:6])
