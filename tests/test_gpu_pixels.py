"""
Parity of the HIP pixel kernels with the oracle, through the C-ABI (sfx_*), on the same seeded inputs.
Tolerance of the path (BASELINE.json north_star): rendered pixels within 1 LSB per channel after 8-bit
quantisation. The generic kernels evaluate the same binary32 operation sequences as the oracle, so they are
additionally REQUIRED to be bit-exact here; the fused and LDS-tiled kernels re-associate and are
held to the 1 LSB bound.
"""
import math

import numpy as np
import pytest

from oracle import binding as O
from shaderflow_amd import _native as N
from tests.helpers import Gpu, gpu_bind_all, lsb_report, oracle_textures, visualizer_inputs

pytestmark = pytest.mark.gpu

ONE_LSB = 1


@pytest.fixture()
def gpu():
    g = Gpu()
    yield g
    g.close()


def assert_within_lsb(got, want, bound=ONE_LSB):
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= bound, lsb_report(got, want)


@pytest.mark.parametrize("name,size", [("default", (256, 256)), ("default", (160, 90)), ("missing", (64, 36)),
                                       ("shadertoy", (80, 45)), ("multi_child", (64, 36)), ("audio", (16, 8))])
def test_untextured_fragments_bit_exact(gpu, name, size):
    w, h = size
    u = O.default_uniforms(w, h, iTime=0.75, iTau=0.075, iAudioVolume=0.4)
    want = O.render(name, u, {}, w, h, threads=4)
    prog, fallback = gpu.program(name)
    assert not fallback
    gpu.set_uniforms(prog, u)
    got = gpu.render(prog, w, h)
    assert np.array_equal(got, want), lsb_report(got, want)


def test_unknown_source_falls_back_to_missing(gpu):
    prog, fallback = gpu.program("void main() { fragColor = vec4(1, 0, 1, 1); /* never seen */ }")
    assert fallback
    w, h = 32, 18
    u = O.default_uniforms(w, h, iTime=3.0)
    gpu.set_uniforms(prog, u)
    assert np.array_equal(gpu.render(prog, w, h), O.render("missing", u, {}, w, h))


def test_non_default_camera_bit_exact(gpu):
    """camera.glsl: stereoscopic and equirectangular projections, moved/zoomed camera"""
    w, h = 96, 54
    for projection in (0, 1, 2):
        u = O.default_uniforms(w, h, iTau=0.3, iCameraProjection=projection, iCameraZoom=1.3, iCameraIsometric=0.2,
                               iCameraPosition=(0.1, -0.05, 0.0), iCameraSeparation=0.07)
        want = O.render("default", u, {}, w, h, threads=4)
        prog, _ = gpu.program("default")
        gpu.set_uniforms(prog, u)
        got = gpu.render(prog, w, h)
        assert np.array_equal(got, want), (projection, lsb_report(got, want))


@pytest.mark.parametrize("name", ["bars", "waveform"])
def test_audio_texture_fragments_bit_exact(gpu, name):
    w, h = 128, 72
    u, arrays, params = visualizer_inputs(w, h, seed=5)
    arrays["iSpectrogram"] = arrays["iSpectrogram"]*3
    want = O.render(name, u, oracle_textures(arrays, params), w, h, threads=4)
    prog, _ = gpu.program(name)
    gpu.set_uniforms(prog, u)
    for key in ("iSpectrogram", "iWaveform"):
        gpu.bind(prog, key + "0x0", gpu.texture(arrays[key], *params[key]))     # sampler names as texture.py:346-347 yields them
    got = gpu.render(prog, w, h)
    assert np.array_equal(got, want), lsb_report(got, want)


def test_multishader_two_passes_bit_exact(gpu):
    w, h = 64, 36
    u = O.default_uniforms(w, h)
    child_o = O.render("multi_child", u, {}, w, h)
    want = O.render("multi_main", u, {"child": O.make_texture(child_o, "linear", True, True)}, w, h)
    child, _ = gpu.program("multi_child")
    gpu.set_uniforms(child, u)
    target = gpu.empty(w, h, 4)
    from shaderflow_amd import _native as N
    N.check(gpu.lib.sfx_render(child, target, 0))
    main, _ = gpu.program("multi_main")
    gpu.set_uniforms(main, u)
    assert gpu.bind(main, "child", target)
    assert np.array_equal(gpu.render(main, w, h), want)


def test_dynamics_scene_user_uniform(gpu):
    w, h = 64, 36
    u, arrays, params = visualizer_inputs(w, h, seed=9)
    u.user[0] = 0.35
    want = O.render("dynamics", u, oracle_textures(arrays, params), w, h)
    prog, _ = gpu.program("dynamics")
    gpu.set_uniforms(prog, u)
    assert gpu.set_float(prog, "iShaderDynamics", 0.35)
    assert not gpu.set_float(prog, "iNotAUniform", 1.0)          # inactive uniforms are ignored, not an error
    gpu.bind(prog, "background", gpu.texture(arrays["background"], *params["background"]))
    assert np.array_equal(gpu.render(prog, w, h), want)


@pytest.mark.parametrize("ssaa,subsample", [(1, 1), (1, 2), (2, 2), (2, 1), (4, 2), (4, 4), (3, 3), (3, 2)])
def test_resolve_pass_bit_exact(gpu, ssaa, subsample):
    rng = np.random.default_rng(ssaa*10 + subsample)
    w, h = 40, 24
    screen = rng.integers(0, 256, (h*ssaa, w*ssaa, 4), dtype=np.uint8)
    want = O.resolve(screen, w, h, subsample)
    got = gpu.resolve(screen, w, h, subsample)
    assert np.array_equal(got, want), lsb_report(got, want)


@pytest.mark.parametrize("w,h", [(40, 24), (41, 23), (64, 17), (130, 50), (255, 9), (17, 5), (200, 16)])
def test_two_pass_tent_kernel_at_every_edge(gpu, w, h):
    """k_resolve_tent (iScreen the size of the output, kernel 2: four pixels per thread, blocks of 64 x 16): heights that are not
    multiples of four or sixteen, widths that are not multiples of 64, frames smaller than a block — bit-exact against final.glsl"""
    rng = np.random.default_rng(w*1000 + h)
    screen = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    got = gpu.resolve(screen, w, h, 2)
    want = O.resolve(screen, w, h, 2)
    assert np.array_equal(got, want), lsb_report(got, want)


def test_resolve_fractional_ssaa(gpu):
    """scene.ssaa = 1.5 → render resolution int(w*1.5) (scene.py:372-375): only the two-pass path applies"""
    rng = np.random.default_rng(3)
    w, h = 40, 24
    screen = rng.integers(0, 256, (int(h*1.5), int(w*1.5), 4), dtype=np.uint8)
    assert np.array_equal(gpu.resolve(screen, w, h, 2), O.resolve(screen, w, h, 2))
    assert gpu.lib.sfx_fused_supported(1500, 2) == 0 and gpu.lib.sfx_fused_supported(2000, 2) == 1


@pytest.mark.parametrize("volume", [0.0, 0.35, 0.8, 1.6])
def test_visualizer_generic_kernel_bit_exact(gpu, volume):
    """PlainShader<visualizer>: float background texture keeps the LDS tile path out"""
    w, h = 96, 54
    u, arrays, params = visualizer_inputs(w, h, seed=11, volume=volume, bg_size=(48, 27))
    arrays["background"] = (arrays["background"].astype(np.float32)/255.0).astype(np.float32)
    want = O.render("visualizer", u, oracle_textures(arrays, params), w, h, threads=8)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    got = gpu.render(prog, w, h)
    assert np.array_equal(got, want), lsb_report(got, want)


@pytest.mark.parametrize("volume,repeat", [(0.0, True), (0.5, True), (0.9, False), (1.7, True)])
def test_visualizer_lds_tile_kernel_within_one_lsb(gpu, volume, repeat):
    """VisualizerShader (uint8 background): blur from the LDS tile, 81 re-associated taps"""
    w, h = 160, 90
    u, arrays, params = visualizer_inputs(w, h, seed=21, volume=volume, bg_size=(120, 68))
    params["background"] = ("linear", repeat, repeat)
    want = O.render("visualizer", u, oracle_textures(arrays, params), w, h, threads=8)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    got = gpu.render(prog, w, h)
    assert_within_lsb(got, want)
    assert (got != want).mean() < 0.02, lsb_report(got, want)


@pytest.mark.parametrize("ssaa", [1, 2])
@pytest.mark.parametrize("camera", [dict(iCameraZoom=1.3, iCameraPosition=(0.12, -0.05, 0.0)), dict(iCameraIsometric=0.3, iCameraZoom=0.8),
                                    dict(iCameraProjection=1, iCameraSeparation=0.08)])
def test_visualizer_lds_tile_kernel_with_a_moved_camera(gpu, camera, ssaa):
    """Non-identity cameras take the reduction path for the tap window (the corner-sample shortcut needs iCamera.gluv == gluv)"""
    w, h = 136, 72
    u, arrays, params = visualizer_inputs(w, h, seed=23, volume=0.8, bg_size=(120, 68))
    for key, value in camera.items():
        cur = getattr(u, key)
        if hasattr(cur, "__len__"):
            for i, v in enumerate(value):
                cur[i] = v
        else:
            setattr(u, key, value)
    u.iSSAA = float(ssaa)
    screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, threads=8)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    if ssaa == 1:
        assert_within_lsb(gpu.render(prog, w, h), screen)
    else:
        assert_within_lsb(gpu.render_resolve(prog, w, h, ssaa, 2), O.resolve(screen, w, h, 2))


@pytest.mark.parametrize("camera", [dict(), dict(iCameraZoom=0.8), dict(iCameraZoom=1.3, iCameraPosition=(0.11, -0.07, 0.0)),
                                    dict(iCameraIsometric=0.35, iCameraZoom=0.9), dict(iCameraDolly=0.4, iCameraFocalLength=1.2),
                                    dict(iCameraPosition=(0.0, 0.0, 1.5))])
@pytest.mark.parametrize("ssaa,size", [(2, (1280, 720)), (2, (1920, 1080)), (4, (640, 360))])
def test_pixel_tier_of_the_strip_kernels_under_axis_cameras(gpu, camera, ssaa, size):
    """Round 6's pixel tier (visualizer.frag:36-62's position-only gains once per output pixel, for the wave tiles k_visualizer_classify
    clears) on the OTHER strip instances and under cameras that zoom and pan: the classification bounds a tile's rectangle through
    camera_along_axis, the pixel tables hold iCamera.gluv of the pixel centres. A spectrogram column whose neighbouring bars differ
    little, so that the tier has tiles (the white-noise column of the test above sends nearly all of them to the per-sample path);
    whole frames against the oracle, and some tile on each path — except from behind the plane, where every fragment is out of bounds."""
    from tests.helpers import smooth_spectrum
    w, h = size
    u, arrays, params = visualizer_inputs(w, h, seed=31, volume=0.8, bg_size=(384, 216))
    arrays["iSpectrogram"] = smooth_spectrum(seed=31)
    for key, value in camera.items():
        cur = getattr(u, key)
        if hasattr(cur, "__len__"):
            for i, v in enumerate(value):
                cur[i] = v
        else:
            setattr(u, key, value)
    u.iSSAA = float(ssaa)
    screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, threads=16)
    want = O.resolve(screen, w, h, 2)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    gpu.ctx.tile_misses()
    got = gpu.render_resolve(prog, w, h, ssaa, 2)
    per_sample_waves = gpu.ctx.tile_misses()
    assert _last_kernel(gpu).startswith("k_visualizer_strip<"), _last_kernel(gpu)
    assert_within_lsb(got, want)
    behind = camera.get("iCameraPosition", (0, 0, 0))[2] > 1.0
    assert per_sample_waves > 0
    if not behind:
        # (the kernel's name carries its geometry: <pitch, rows, S, WALK, waves, column groups, half>: a wave is 64 columns x WALK rows)
        walk = int(_last_kernel(gpu).split("<")[1].split(",")[3])
        waves = -(-w*ssaa//64)*(-(-h*ssaa//walk))
        assert per_sample_waves < 0.9*waves, (per_sample_waves, waves, _last_kernel(gpu))


@pytest.mark.parametrize("camera", [dict(iCameraZoom=0.8), dict(iCameraZoom=1.3, iCameraPosition=(0.11, -0.07, 0.0)),
                                    dict(iCameraIsometric=0.35, iCameraZoom=0.9), dict(iCameraDolly=0.4, iCameraFocalLength=1.2),
                                    dict(iCameraPosition=(0.0, 0.0, 1.5))])
@pytest.mark.parametrize("ssaa", [1, 2, 4])
def test_strip_kernel_with_a_zoomed_or_panned_camera(gpu, camera, ssaa):
    """A perspective camera with the untouched basis (zoom, pan, isometric factor, dolly, focal length — no rotation) keeps
    iCamera.gluv.x a function of the sample column and .y of the row (glsl.hpp camera_is_axis_aligned), so the strip kernel's tables
    take get_camera per axis: whole frames against the oracle, the strip kernel asserted. The last camera looks at the plane from
    behind (t < 0): every fragment is out of bounds."""
    w, h = (640, 360) if ssaa < 4 else (320, 180)
    u, arrays, params = visualizer_inputs(w, h, seed=29, volume=0.8, bg_size=(384, 216))
    for key, value in camera.items():
        cur = getattr(u, key)
        if hasattr(cur, "__len__"):
            for i, v in enumerate(value):
                cur[i] = v
        else:
            setattr(u, key, value)
    u.iSSAA = float(ssaa)
    screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, threads=8)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    if ssaa == 1:
        got, want = gpu.render(prog, w, h), screen
    else:
        got, want = gpu.render_resolve(prog, w, h, ssaa, 2), O.resolve(screen, w, h, 2)
    assert _last_kernel(gpu).startswith("k_visualizer_strip<"), _last_kernel(gpu)
    assert_within_lsb(got, want)


def _turn_camera(u, roll=0.0, tilt=0.0, **others):
    """The camera basis rolled about its forward axis, then tilted about its right axis (degrees) — what camera.py's rotate / rotate2d
    leave in iCameraRight / iCameraUpward / iCameraForward"""
    c, s = math.cos(math.radians(roll)), math.sin(math.radians(roll))
    ct, st = math.cos(math.radians(tilt)), math.sin(math.radians(tilt))
    right, up, forward = (c, s, 0.0), (-s*ct, c*ct, st), (s*st, -c*st, ct)
    for name, vector in (("iCameraRight", right), ("iCameraUpward", up), ("iCameraForward", forward)):
        for i, v in enumerate(vector):
            getattr(u, name)[i] = v
    for key, value in others.items():
        cur = getattr(u, key)
        if hasattr(cur, "__len__"):
            for i, v in enumerate(value):
                cur[i] = v
        else:
            setattr(u, key, value)


@pytest.mark.parametrize("camera,all_tiled", [(dict(roll=17.0), True), (dict(roll=45.0), True), (dict(roll=90.0), True),
                                              (dict(roll=-30.0, iCameraZoom=1.25, iCameraPosition=(0.1, -0.06, 0.0)), True),
                                              (dict(roll=200.0, iCameraIsometric=0.3), True), (dict(roll=10.0, tilt=12.0), False)])
def test_visualizer_lds_tile_kernel_with_a_rolled_camera(gpu, camera, all_tiled):
    """A camera rolled about its forward axis mixes the screen axes: a block's tap window is bounded from the camera's four slopes
    (capi camera_slopes), the tile is sized per launch and the block shape is the one that stages the fewest cells per pixel. With
    the affine (rolled, zoomed, panned) cameras no block may fall back to the generic taps; a tilted camera is projective and its
    bound is a heuristic, so only the pixels are asserted."""
    w, h, ssaa = 640, 360, 2
    u, arrays, params = visualizer_inputs(w, h, seed=31, volume=0.8, bg_size=(384, 216))
    _turn_camera(u, **camera)
    u.iSSAA = float(ssaa)
    screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, threads=8)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    gpu.ctx.tile_misses()
    got = gpu.render_resolve(prog, w, h, ssaa, 2)
    misses = gpu.ctx.tile_misses()
    assert _last_kernel(gpu).startswith("k_render_resolve<VisualizerShader<0, 0, "), _last_kernel(gpu)
    if all_tiled:
        assert misses == 0, f"{misses} blocks ran the generic taps"
    assert_within_lsb(got, O.resolve(screen, w, h, 2))


@pytest.mark.parametrize("bg_size,ssaa", [((160, 90), 2), ((320, 180), 2), ((640, 360), 2), ((320, 180), 4)])
def test_visualizer_dense_backgrounds_use_a_tile_sized_per_launch(gpu, bg_size, ssaa):
    """Backgrounds with more texels per shaded sample than the fixed tile holds: dynamic-LDS tile, narrower blocks as needed"""
    w, h = 200, 48
    u, arrays, params = visualizer_inputs(w, h, seed=61, volume=0.9, bg_size=bg_size)
    u.iSSAA = float(ssaa)
    screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, threads=8)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    assert_within_lsb(gpu.render_resolve(prog, w, h, ssaa, 2), O.resolve(screen, w, h, 2))


def test_visualizer_tile_overflow_falls_back(gpu):
    """A background far larger than the output: the tap window exceeds the LDS tile → generic taps, same result"""
    w, h = 64, 36
    u, arrays, params = visualizer_inputs(w, h, seed=4, volume=1.2, bg_size=(1536, 864))
    want = O.render("visualizer", u, oracle_textures(arrays, params), w, h, threads=8)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    assert_within_lsb(gpu.render(prog, w, h), want)


@pytest.mark.parametrize("name", ["default", "visualizer", "bars"])
@pytest.mark.parametrize("ssaa,subsample", [(1, 1), (2, 2), (2, 1), (4, 2), (4, 4)])
def test_fused_render_resolve_within_one_lsb(gpu, name, ssaa, subsample):
    """sfx_render_resolve vs the oracle's two passes (render at w*ssaa, RGBA8, then final.glsl)"""
    w, h = 136, 40                                                # not a multiple of the 128-pixel block
    u, arrays, params = visualizer_inputs(w, h, seed=31, volume=0.7, bg_size=(100, 56))
    u.iSSAA = float(ssaa)
    screen = O.render(name, u, oracle_textures(arrays, params), w*ssaa, h*ssaa, threads=8)
    want = O.resolve(screen, w, h, subsample)
    prog, _ = gpu.program(name)
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    got = gpu.render_resolve(prog, w, h, ssaa, subsample)
    assert_within_lsb(got, want)


@pytest.mark.parametrize("camera", [dict(), dict(iCameraZoom=1.4, iCameraPosition=(0.2, -0.1, 0.0)), dict(iCameraIsometric=0.3, iCameraDolly=0.5)])
@pytest.mark.parametrize("w,h,tau", [(136, 40, 0.0), (75, 75, 0.31), (301, 169, 0.77), (640, 360, 0.5)])
def test_default_fragment_separable_kernel_within_one_lsb(gpu, w, h, tau, camera):
    """default.glsl at 2x SSAA under the identity camera runs k_separable_fused<default> (per-column / per-row tables, hardware
    reciprocals and logarithms for its colour-only polar terms): whole frames against the oracle, odd sizes included (a sample lands
    on gluv = (0, 0)), several iTau (the hue shift), both final.glsl kernels the fused path serves"""
    u, arrays, params = visualizer_inputs(w, h, seed=3)
    u.iSSAA = 2.0
    u.iTau = tau
    for key, value in camera.items():                                  # zoomed / panned cameras stay separable per axis
        cur = getattr(u, key)
        if hasattr(cur, "__len__"):
            for i, v in enumerate(value):
                cur[i] = v
        else:
            setattr(u, key, value)
    prog, _ = gpu.program("default")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    screen = O.render("default", u, oracle_textures(arrays, params), w*2, h*2, threads=8)
    for subsample in (2, 1):
        got = gpu.render_resolve(prog, w, h, 2, subsample)
        assert _last_kernel(gpu) == "k_separable_fused<default>", _last_kernel(gpu)
        assert_within_lsb(got, O.resolve(screen, w, h, subsample))


def test_fused_matches_two_pass_on_device(gpu):
    """Same device, same inputs: sfx_render + sfx_resolve vs sfx_render_resolve"""
    from shaderflow_amd import _native as N
    w, h, ssaa = 200, 64, 2
    u, arrays, params = visualizer_inputs(w, h, seed=41, volume=1.0)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    screen = gpu.empty(w*ssaa, h*ssaa, 4)
    N.check(gpu.lib.sfx_texture_params(screen, 1, 0, 0))
    N.check(gpu.lib.sfx_render(prog, screen, 0))
    final = gpu.empty(w, h, 3)
    N.check(gpu.lib.sfx_resolve(gpu.ctx.handle, screen, final, 2))
    two_pass = gpu.read(final, w, h, 3)
    fused = gpu.render_resolve(prog, w, h, ssaa, 2)
    assert_within_lsb(fused, two_pass)


def test_unsupported_fused_pair_is_refused(gpu):
    from shaderflow_amd import _native as N
    prog, _ = gpu.program("default")
    dst = gpu.empty(32, 18, 3)
    assert gpu.lib.sfx_render_resolve(prog, dst, 1, 2) == N.E_UNSUPPORTED       # 3x3 tent leaves the pixel
    assert b"footprint" in gpu.lib.sfx_last_error()


def test_full_size_properties_4k_ssaa2(gpu):
    """BASELINE config 3 size (3840x2160, 2xSSAA): size-independent properties instead of a CPU render:
    (a) a band of rows equals the oracle's band, (b) horizontal translation invariance of `bars` columns,
    (c) determinism (two launches give identical frames)."""
    w, h, ssaa = 3840, 2160, 2
    u, arrays, params = visualizer_inputs(w, h, seed=51, volume=0.9, bg_size=(1920, 1080))
    u.iSSAA = 2.0
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    a = gpu.render_resolve(prog, w, h, ssaa, 2)
    assert gpu.lib.sfx_last_kernel().decode().startswith("k_visualizer_"), gpu.lib.sfx_last_kernel()     # the bench's kernel
    b = gpu.render_resolve(prog, w, h, ssaa, 2)
    assert np.array_equal(a, b)
    rows = (1000, 1004)                                           # 4 output rows = 8 supersample rows on the CPU
    screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, rows=(rows[0]*ssaa, rows[1]*ssaa), threads=8)
    want = O.resolve(screen, w, h, 2, rows=rows, threads=8)
    assert_within_lsb(a[rows[0]:rows[1]], want[rows[0]:rows[1]])
    # bars.frag depends on astuv only through texture(iSpectrogram, astuv.yx): columns inside one spectrogram bin are equal
    bars, _ = gpu.program("bars")
    gpu.set_uniforms(bars, u)
    gpu.bind(bars, "iSpectrogram", gpu.texture(arrays["iSpectrogram"], *params["iSpectrogram"]))
    img = gpu.render_resolve(bars, w, h, ssaa, 2)
    bin_width = w/115
    assert np.array_equal(img[:, int(10*bin_width) + 3], img[:, int(10*bin_width) + 20])


def _last_kernel(gpu) -> str:
    return gpu.lib.sfx_last_kernel().decode()


@pytest.mark.parametrize("w,h,volume", [(600, 362, 0.9), (640, 361, 0.0), (520, 363, 1.3)])
def test_strip_kernel_partial_blocks_whole_frame_against_oracle(gpu, w, h, volume):
    """The benchmark's kernel (k_visualizer_strip: per-frame column/row tables, lanes walking strips of four samples of their column)
    on frames whose width is not a multiple of its 128-pixel blocks and whose height is not a multiple of its four rows, loud and
    silent (blur radius 0: every line weight in one cell): the WHOLE frame against the oracle, and twice for determinism"""
    ssaa = 2
    u, arrays, params = visualizer_inputs(w, h, seed=61, volume=volume, bg_size=(384, 216))
    u.iSSAA = 2.0
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    got = gpu.render_resolve(prog, w, h, ssaa, 2)
    assert gpu.lib.sfx_last_kernel().decode().startswith("k_visualizer_strip<"), gpu.lib.sfx_last_kernel()
    assert np.array_equal(got, gpu.render_resolve(prog, w, h, ssaa, 2))
    screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, threads=8)
    assert_within_lsb(got, O.resolve(screen, w, h, 2, threads=8))


def test_full_size_properties_1080p_two_pass(gpu):
    """BASELINE config 2 at its own size (1920x1080, ssaa 1, subsample 2): the fragment into an RGBA8 iScreen, then final.glsl's 3x3
    tent as a pass — the pair the fused kernel refuses. Kernel selection depends on the size (window bound of the LDS tile), so the
    instance is asserted; a band of rows of BOTH passes equals the oracle; two launches give identical frames."""
    w, h = 1920, 1080
    u, arrays, params = visualizer_inputs(w, h, seed=52, volume=0.9, bg_size=(1920, 1080))
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    assert gpu.lib.sfx_fused_supported(1000, 2) == 0
    screen = gpu.empty(w, h, 4)
    N.check(gpu.lib.sfx_texture_params(screen, 1, 0, 0))
    N.check(gpu.lib.sfx_render(prog, screen, 0))
    assert _last_kernel(gpu).startswith("k_visualizer_strip<66, 22, 1, 2, "), _last_kernel(gpu)
    shaded = gpu.read(screen, w, h, 4)
    final = gpu.empty(w, h, 3)
    N.check(gpu.lib.sfx_resolve(gpu.ctx.handle, screen, final, 2))
    frame = gpu.read(final, w, h, 3)
    N.check(gpu.lib.sfx_render(prog, screen, 0))
    N.check(gpu.lib.sfx_resolve(gpu.ctx.handle, screen, final, 2))
    assert np.array_equal(frame, gpu.read(final, w, h, 3))
    # the tent reaches one row up and down: shade rows [r0-1, r1+1) on the CPU, compare the resolve of the inner rows
    for r0, r1 in ((0, 3), (537, 543), (1077, 1080)):
        lo, hi = max(0, r0 - 1), min(h, r1 + 1)
        want_screen = O.render("visualizer", u, oracle_textures(arrays, params), w, h, rows=(lo, hi), threads=8)
        assert_within_lsb(shaded[lo:hi], want_screen[lo:hi])
        # final.glsl over the DEVICE's own iScreen rows is bit-exact (k_resolve is the generic chain)
        padded = shaded.copy()
        want = O.resolve(padded, w, h, 2, rows=(r0, r1), threads=8)
        assert np.array_equal(frame[r0:r1], want[r0:r1])


def test_full_size_properties_8k_ssaa4(gpu):
    """BASELINE config 4 at its own size (7680x4320, 4x SSAA: 30720x17280 samples, never materialised): the kernel instance, a band
    of rows against the oracle (16 supersamples per pixel resolved by final.glsl with subsample 2), determinism."""
    w, h, ssaa = 7680, 4320, 4
    u, arrays, params = visualizer_inputs(w, h, seed=53, volume=0.9, bg_size=(1920, 1080))
    u.iSSAA = 4.0
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    a = gpu.render_resolve(prog, w, h, ssaa, 2)
    assert _last_kernel(gpu).startswith("k_visualizer_strip<20, 13, 4, "), _last_kernel(gpu)      # (the 20-cell pitch of round 6's tile sweep: 17 cells of window)
    b = gpu.render_resolve(prog, w, h, ssaa, 2)
    assert np.array_equal(a, b)
    for rows in ((0, 1), (2159, 2161), (4319, 4320)):
        screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, rows=(rows[0]*ssaa, rows[1]*ssaa), threads=8)
        want = O.resolve(screen, w, h, 2, rows=rows, threads=8)
        assert_within_lsb(a[rows[0]:rows[1]], want[rows[0]:rows[1]])


@pytest.mark.parametrize("w,h,ssaa,kernel", [(1920, 1080, 2, "k_visualizer_strip<120, 13, 2, 6, "), (2560, 1440, 2, "k_visualizer_strip<120, 13, 2, 6, "),
                                             (1920, 1080, 4, "k_visualizer_strip<40, 14, 4, 6, "), (1280, 720, 2, "k_visualizer_strip<92, 16, 2, 3, "),
                                             (1280, 720, 4, "k_visualizer_strip<56, 16, 4, 6, ")])
def test_full_size_properties_dense_outputs(gpu, w, h, ssaa, kernel):
    """Outputs denser than the benchmark's over the same 1080-row background (up to 0.43 texel per sample): the strip kernel's
    instances with shorter strips over larger tiles; bands of rows against the oracle, determinism."""
    u, arrays, params = visualizer_inputs(w, h, seed=59, volume=0.9, bg_size=(1920, 1080))
    u.iSSAA = float(ssaa)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    a = gpu.render_resolve(prog, w, h, ssaa, 2)
    assert _last_kernel(gpu).startswith(kernel), _last_kernel(gpu)
    assert np.array_equal(a, gpu.render_resolve(prog, w, h, ssaa, 2))
    for rows in ((0, 2), (h//2 - 1, h//2 + 1), (h - 1, h)):
        screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, rows=(rows[0]*ssaa, rows[1]*ssaa), threads=8)
        want = O.resolve(screen, w, h, 2, rows=rows, threads=8)
        assert_within_lsb(a[rows[0]:rows[1]], want[rows[0]:rows[1]])


def test_destroyed_texture_is_unbound_not_dangling(gpu):
    """A program must not keep a pointer to a texture that was destroyed: the render reports the missing sampler instead"""
    from shaderflow_amd import _native as N
    import ctypes as C
    prog, _ = gpu.program("dynamics")
    gpu.set_uniforms(prog, O.default_uniforms(32, 18))
    texture = N.Handle()
    N.check(gpu.lib.sfx_texture_create(gpu.ctx.handle, 8, 8, 3, N.U8, C.byref(texture)))
    assert gpu.bind(prog, "background", texture)
    target = gpu.empty(32, 18, 4)
    assert gpu.lib.sfx_render(prog, target, 0) == N.OK
    N.check(gpu.lib.sfx_texture_destroy(texture))
    assert gpu.lib.sfx_render(prog, target, 0) != N.OK and b"background" in gpu.lib.sfx_last_error()


def test_float16_textures_and_render_targets_bit_exact(gpu):
    """numpy float16 textures ("f2", texture.py:28-38): sampled (nearest and linear) and used as a render target"""
    rng = np.random.default_rng(8)
    data = (rng.standard_normal((27, 48, 3))*3).astype(np.float16)
    w, h = 96, 54
    u = O.default_uniforms(w, h, iCameraZoom=0.8)
    for filter in ("nearest", "linear"):
        want = O.render("video", u, {0: O.make_texture(data, filter, True, True)}, w, h)
        prog, _ = gpu.program("video")
        gpu.set_uniforms(prog, u)
        assert gpu.bind(prog, "iVideo", gpu.texture(data, filter, True, True))
        assert np.array_equal(gpu.render(prog, w, h), want)
    wide = O.render_to("video", u, {0: O.make_texture(data, "linear", True, True)}, w, h, 4, np.float16)
    got = gpu.render(prog, w, h, comps=4, dtype=np.float16)
    assert np.array_equal(got.view(np.uint16), wide.view(np.uint16))


def same_as_generic(name: str, fused: np.ndarray, generic: np.ndarray, runs: bool) -> None:
    """The per-pixel kernel reproduces the generic fused kernel byte for byte. The run kernel does for red and green and for all of
    waveform; its blue is the INTEGER mean of the four sample bytes, (sum + 2) >> 2, where resolve_channel rounds a float mean: they
    differ, by 1 LSB, only where the four bytes sum to 2 (mod 4) — a tie the float chain's rounding noise decides"""
    if not runs or name == "waveform":
        assert np.array_equal(fused, generic), lsb_report(fused, generic)
        return
    assert np.array_equal(fused[..., :2], generic[..., :2]), lsb_report(fused[..., :2], generic[..., :2])
    d = np.abs(fused[..., 2].astype(int) - generic[..., 2].astype(int))
    assert d.max() <= 1 and (d == 0).mean() >= 0.7, lsb_report(fused[..., 2], generic[..., 2])


@pytest.mark.parametrize("name", ["bars", "waveform"])
@pytest.mark.parametrize("w,h,subsample,gain", [(600, 362, 2, 3.0), (1279, 717, 1, 40.0), (257, 33, 2, 0.2), (1280, 720, 1, 3.0), (644, 130, 2, 0.5)])
def test_separable_audio_fragments_equal_the_generic_fused_kernel(gpu, name, w, h, subsample, gain, monkeypatch):
    """k_separable_runs (widths that are multiples of 4: rows as runs, four pixels = one 12-byte store per lane) and
    k_separable_fused<bars|waveform> (any width: per-pixel row counts) against the generic fused kernel shading every supersample
    (SHADERFLOW_SEPARABLE=0), on odd sizes, partial blocks in both directions, both resolve kernels, with bars from far below the
    frame to far above it and non-finite spectrogram values (negative power: sqrt gives NaN) — and within 1 LSB of render + resolve"""
    u, arrays, params = visualizer_inputs(w, h, seed=77, volume=0.6)
    u.iSSAA = 2.0
    spectrogram = arrays["iSpectrogram"].astype(np.float32)*gain
    flat = spectrogram.reshape(-1)
    flat[3] = -1.0; flat[10] = np.inf; flat[17] = np.nan; flat[24] = 0.0; flat[31] = 120.0**2; flat[38] = np.float32(120.0**2)*np.float32(0.25)
    arrays["iSpectrogram"] = spectrogram
    prog, _ = gpu.program(name)
    gpu.set_uniforms(prog, u)
    for key in ("iSpectrogram", "iWaveform"):
        gpu.bind(prog, key + "0x0", gpu.texture(arrays[key], *params[key]))
    fused = gpu.render_resolve(prog, w, h, 2, subsample)
    runs = w % 4 == 0
    assert _last_kernel(gpu) == (f"k_separable_runs<{name}>" if runs else f"k_separable_fused<{name}>"), _last_kernel(gpu)
    top_down = None
    if runs:
        monkeypatch.setenv("SHADERFLOW_SEPARABLE_RUNS", "0")
        per_pixel = gpu.render_resolve(prog, w, h, 2, subsample)
        assert _last_kernel(gpu) == f"k_separable_fused<{name}>", _last_kernel(gpu)
        same_as_generic(name, fused, per_pixel, True)
    monkeypatch.setenv("SHADERFLOW_SEPARABLE", "0")
    generic = gpu.render_resolve(prog, w, h, 2, subsample)
    assert _last_kernel(gpu).startswith("k_render_resolve<"), _last_kernel(gpu)
    same_as_generic(name, fused, generic, runs)
    two_pass = gpu.resolve(gpu.render(prog, 2*w, 2*h), w, h, subsample)
    assert_within_lsb(fused, two_pass)


@pytest.mark.parametrize("name", ["bars", "waveform"])
def test_separable_audio_fragments_at_8k_where_blocks_walk_32_rows(gpu, name, monkeypatch):
    """One 7680x4320 frame at 2x, whole frame against the generic fused kernel: the run kernel (blocks of 256 pixels x 128 rows) and,
    with SHADERFLOW_SEPARABLE_RUNS=0, the 32-rows-per-block instance of k_separable_fused that launches of more than 2 048 blocks take"""
    w, h = 7680, 4320
    u, arrays, params = visualizer_inputs(w, h, seed=78, volume=0.8)
    u.iSSAA = 2.0
    prog, _ = gpu.program(name)
    gpu.set_uniforms(prog, u)
    for key in ("iSpectrogram", "iWaveform"):
        gpu.bind(prog, key + "0x0", gpu.texture(arrays[key], *params[key]))
    runs = gpu.render_resolve(prog, w, h, 2, 2)
    assert _last_kernel(gpu) == f"k_separable_runs<{name}>", _last_kernel(gpu)
    monkeypatch.setenv("SHADERFLOW_SEPARABLE_RUNS", "0")
    fused = gpu.render_resolve(prog, w, h, 2, 2)
    assert _last_kernel(gpu) == f"k_separable_fused<{name}>", _last_kernel(gpu)
    monkeypatch.setenv("SHADERFLOW_SEPARABLE", "0")
    generic = gpu.render_resolve(prog, w, h, 2, 2)
    assert _last_kernel(gpu).startswith("k_render_resolve<"), _last_kernel(gpu)
    assert np.array_equal(fused, generic)
    same_as_generic(name, runs, generic, True)
    assert len(np.unique(fused[::16, ::16].reshape(-1, 3), axis=0)) > 4     # not a blank frame


@pytest.mark.parametrize("name", ["default", "bars", "mandelbrot"])
@pytest.mark.parametrize("ssaa,subsample", [(1, 1), (2, 2), (2, 1), (4, 4)])
def test_generic_fused_kernel_on_frames_whose_rows_leave_in_one_sweep(gpu, name, ssaa, subsample, monkeypatch):
    """k_render_resolve (the generic fused kernel: one thread or quad per output pixel) on a width of whole 128-pixel blocks and whole
    16-byte groups — the path where a block's rows leave in one sweep of 16-byte stores (other widths store row by row and are
    covered above) — against the oracle's two passes, with a height that leaves the last block partial"""
    monkeypatch.setenv("SHADERFLOW_SEPARABLE", "0")                # default / bars: the generic kernel instead of the separable one
    w, h = 256, 37
    u, arrays, params = visualizer_inputs(w, h, seed=33, volume=0.7, bg_size=(100, 56))
    u.iSSAA = float(ssaa)
    screen = O.render(name, u, oracle_textures(arrays, params), w*ssaa, h*ssaa, threads=8)
    want = O.resolve(screen, w, h, subsample)
    prog, _ = gpu.program(name)
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    got = gpu.render_resolve(prog, w, h, ssaa, subsample)
    assert _last_kernel(gpu).startswith("k_render_resolve<"), _last_kernel(gpu)
    assert_within_lsb(got, want)
    again = gpu.render_resolve(prog, w, h, ssaa, subsample)        # determinism of the sweep
    assert np.array_equal(got, again)
