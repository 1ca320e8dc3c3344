"""
Seeded differential runs of the HIP kernels against the oracle over odd sizes, cameras, volumes, background shapes and SSAA pairs
— partial blocks, tiles of every kind (fixed, per-launch, narrowed blocks, generic fallback), wrap modes. Bounds as everywhere:
generic kernels bit-exact, the tiled visualizer and the fused resolve within 1 LSB.
"""
import numpy as np
import pytest

from oracle import binding as O
from tests.helpers import Gpu, gpu_bind_all, lsb_report, oracle_textures, visualizer_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    g = Gpu()
    yield g
    g.close()


def random_camera(rng):
    if rng.random() < 0.5:
        return {}
    return dict(iCameraProjection=int(rng.integers(0, 3)), iCameraZoom=float(rng.uniform(0.6, 1.8)), iCameraIsometric=float(rng.uniform(0, 0.4)),
                iCameraPosition=(float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2)), 0.0), iCameraSeparation=float(rng.uniform(0.02, 0.1)))


@pytest.mark.parametrize("seed", range(18))
def test_visualizer_rolled_and_tilted_cameras_random(gpu, seed):
    """Cameras that are neither the identity nor axis aligned (rolled about the forward axis, tilted about the right one, zoomed and
    panned on top) over random sizes and backgrounds, no SSAA / 2x / 4x: the host bounds a block's tap window from the camera's slopes
    (capi camera_slopes) and picks tile and block shape per launch; whatever it picks — and whatever a block that still misses its
    tile falls back to — is within 1 LSB of the oracle."""
    import math
    rng = np.random.default_rng(9300 + seed)
    ssaa = (2, 1, 4)[seed % 3]
    w = int(rng.integers(64, 520 if ssaa < 4 else 260)); h = int(rng.integers(40, 300 if ssaa < 4 else 150))
    bg_size = (int(rng.integers(64, 480)), int(rng.integers(36, 270)))
    u, arrays, params = visualizer_inputs(w, h, seed=seed, volume=float(rng.choice([0.0, 0.5, 0.9, 1.4])), bg_size=bg_size, time=float(rng.uniform(0, 40)))
    params["background"] = ("linear", bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
    roll, tilt = math.radians(float(rng.uniform(-180, 180))), math.radians(float(rng.choice([0.0, 0.0, 8.0, -15.0])))
    c, s_, ct, st = math.cos(roll), math.sin(roll), math.cos(tilt), math.sin(tilt)
    for name, vector in (("iCameraRight", (c, s_, 0.0)), ("iCameraUpward", (-s_*ct, c*ct, st)), ("iCameraForward", (s_*st, -c*st, ct))):
        for i, v in enumerate(vector):
            getattr(u, name)[i] = v
    if seed % 2:
        u.iCameraZoom = float(rng.uniform(0.7, 1.6)); u.iCameraPosition[0] = float(rng.uniform(-0.2, 0.2)); u.iCameraPosition[1] = float(rng.uniform(-0.2, 0.2))
    u.iSSAA = float(ssaa)
    screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, threads=8)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    got, want = (gpu.render(prog, w, h), screen) if ssaa == 1 else (gpu.render_resolve(prog, w, h, ssaa, 2), O.resolve(screen, w, h, 2, threads=4))
    assert not gpu.lib.sfx_last_kernel().decode().startswith("k_visualizer_strip"), gpu.lib.sfx_last_kernel()
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1, (seed, (w, h, ssaa, bg_size), gpu.lib.sfx_last_kernel().decode(), lsb_report(got, want))


@pytest.mark.parametrize("seed", range(48))
def test_default_fragment_random_sizes_and_cameras(gpu, seed):
    """default.glsl's fused kernel (k_separable_fused<default>) resolves most pixels from column and row MEANS and shares one evaluation
    of the polar terms between the rows of a walk, four rows or a pixel — which tier a pixel takes depends on its size in gluv units
    and on where the ring and the checkerboard's edges fall, so: random sizes from thumbnails to 2560x1440 (walks of 8 and of 16 rows,
    partial blocks, widths that are not multiples of four), zoomed / panned / dollied cameras (the ring anywhere, or nowhere), a wanted
    aspect narrower than the frame (columns out of bounds), any iTau, bottom-up and top-down rows; whole frames against the oracle
    within 1 LSB."""
    from shaderflow_amd import _native as N
    rng = np.random.default_rng(9100 + seed)
    w, h = [(int(rng.integers(40, 900)), int(rng.integers(24, 500))), (1280, 720), (1920, 1080), (2560, 1440), (1000, 1000), (3001, 403)][min(seed % 8, 5)]
    u, arrays, params = visualizer_inputs(w, h, seed=seed)
    u.iSSAA, u.iTau = 2.0, float(rng.random())
    if seed % 3:
        u.iCameraZoom = float(rng.choice([0.2, 0.55, 0.74, 1.0, 1.3, 2.2, 5.0]))
        u.iCameraPosition[0] = float(rng.uniform(-0.8, 0.8)); u.iCameraPosition[1] = float(rng.uniform(-0.5, 0.5))
        u.iCameraIsometric = float(rng.choice([0.0, 0.3])); u.iCameraDolly = float(rng.choice([0.0, 0.4]))
    if seed % 5 == 4:
        u.iWantAspect = float(0.8*w/h)
    prog, _ = gpu.program("default")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    top_down = bool(rng.integers(0, 2))
    N.check(gpu.lib.sfx_ctx_output_top_down(gpu.ctx.handle, int(top_down)))
    try:
        got = gpu.render_resolve(prog, w, h, 2, 2)
    finally:
        N.check(gpu.lib.sfx_ctx_output_top_down(gpu.ctx.handle, 0))
    assert gpu.lib.sfx_last_kernel().decode() == "k_separable_fused<default>", gpu.lib.sfx_last_kernel()
    screen = O.render("default", u, oracle_textures(arrays, params), w*2, h*2, threads=8)
    want = O.resolve(screen, w, h, 2, threads=8)
    if top_down:
        want = want[::-1]
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1, (seed, (w, h), lsb_report(got, want), np.argwhere(d > 1)[:5].tolist())


@pytest.mark.parametrize("seed", range(24))
def test_visualizer_random_configurations(gpu, seed):
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(17, 260)), int(rng.integers(9, 120))
    ssaa, subsample = [(1, 1), (2, 2), (2, 1), (4, 2), (4, 4), (1, 2), (3, 2)][seed % 7]
    bg_size = (int(rng.integers(8, 400)), int(rng.integers(8, 260)))
    u, arrays, params = visualizer_inputs(w, h, seed=seed, volume=float(rng.choice([0.0, 0.2, 0.7, 1.1, 2.0])), bg_size=bg_size,
                                          time=float(rng.uniform(0, 30)), std=float(rng.uniform(0, 0.5)))
    params["background"] = ("linear", bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
    for key, value in random_camera(rng).items():
        cur = getattr(u, key)
        if hasattr(cur, "__len__"):
            for i, v in enumerate(value):
                cur[i] = v
        else:
            setattr(u, key, value)
    u.iSSAA = float(ssaa)
    screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, threads=8)
    want = O.resolve(screen, w, h, subsample, threads=4)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    from shaderflow_amd import _native as N
    if N.lib().sfx_fused_supported(ssaa*1000, subsample):
        got = gpu.render_resolve(prog, w, h, ssaa, subsample)
    else:
        shaded = gpu.render(prog, w*ssaa, h*ssaa)
        d = np.abs(shaded.astype(int) - screen.astype(int))
        assert d.max() <= 1, lsb_report(shaded, screen)
        got = gpu.resolve(screen, w, h, subsample)
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1, (seed, (w, h, ssaa, subsample, bg_size), lsb_report(got, want))


def _strip_seeds():
    import os
    return range(int(os.environ.get("SHADERFLOW_FUZZ_STRIP_SEEDS", 18)))


@pytest.mark.parametrize("seed", _strip_seeds())
def test_strip_kernel_random_configurations(gpu, seed):
    """The benchmark's kernel family (k_visualizer_strip: identity camera) over random sizes, backgrounds, wrap modes, volumes
    (blur radius 0 … beyond the line slots), SSAA 2 and 4 fused and no SSAA into an RGBA8 iScreen, bottom-up and top-down rows:
    whole frames against the oracle within 1 LSB. Whatever the launch picks (a strip instance, the quad kernel, round 1's kernels
    when no tile fits) has to agree; most of these sizes land on strip instances with partial blocks and partial strips.
    SHADERFLOW_FUZZ_STRIP_SEEDS=<n> runs more seeds."""
    from shaderflow_amd import _native as N
    rng = np.random.default_rng(7000 + seed)
    ssaa = (2, 4, 1)[seed % 3]
    w = int(rng.integers(40, 700 if ssaa < 4 else 360))
    h = int(rng.integers(24, 420 if ssaa < 4 else 200))
    bg_size = (int(rng.integers(64, 640)), int(rng.integers(36, 360)))
    u, arrays, params = visualizer_inputs(w, h, seed=seed, volume=float(rng.choice([0.0, 0.1, 0.5, 0.9, 1.4, 2.5])), bg_size=bg_size,
                                          time=float(rng.uniform(0, 40)), std=float(rng.uniform(0, 0.6)))
    params["background"] = ("linear", bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
    if rng.random() < 0.4:                                           # a zoomed / panned camera (no rotation): still the tables' case
        u.iCameraZoom = float(rng.uniform(0.6, 1.8)); u.iCameraIsometric = float(rng.uniform(0, 0.4))
        u.iCameraPosition[0] = float(rng.uniform(-0.2, 0.2)); u.iCameraPosition[1] = float(rng.uniform(-0.2, 0.2))
        u.iCameraDolly = float(rng.uniform(0, 0.5)); u.iCameraFocalLength = float(rng.uniform(0.8, 1.5))
    u.iSSAA = float(ssaa)
    screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, threads=8)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    if ssaa == 1:
        got, want = gpu.render(prog, w, h), screen
    else:
        top_down = bool(rng.integers(0, 2))
        subsample = int(rng.choice([s for s in (1, 2, 4) if N.lib().sfx_fused_supported(ssaa*1000, s)]))       # final.glsl's kernel
        N.check(gpu.lib.sfx_ctx_output_top_down(gpu.ctx.handle, int(top_down)))
        try:
            got = gpu.render_resolve(prog, w, h, ssaa, subsample)
        finally:
            N.check(gpu.lib.sfx_ctx_output_top_down(gpu.ctx.handle, 0))
        want = O.resolve(screen, w, h, subsample, threads=4)
        if top_down:
            want = want[::-1]
    import os
    if os.environ.get("SHADERFLOW_FUZZ_VERBOSE"):
        print(f"\nseed {seed}: {w}x{h} ssaa {ssaa} background {bg_size}: {gpu.lib.sfx_last_kernel().decode()}")
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1, (seed, (w, h, ssaa, bg_size), gpu.lib.sfx_last_kernel().decode(), lsb_report(got, want))


@pytest.mark.parametrize("seed", range(16))
def test_generic_fragments_random_configurations(gpu, seed):
    rng = np.random.default_rng(2000 + seed)
    name = ["default", "bars", "waveform", "shadertoy", "raymarch", "mandelbrot", "missing", "tetration"][seed % 8]
    w, h = int(rng.integers(5, 300)), int(rng.integers(3, 150))
    u, arrays, params = visualizer_inputs(w, h, seed=seed, time=float(rng.uniform(0, 20)))
    for key, value in random_camera(rng).items():
        cur = getattr(u, key)
        if hasattr(cur, "__len__"):
            for i, v in enumerate(value):
                cur[i] = v
        else:
            setattr(u, key, value)
    u.iQuality = float(rng.uniform(0.05, 0.3))
    want = O.render(name, u, oracle_textures(arrays, params), w, h, threads=8)
    prog, _ = gpu.program(name)
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, {k: v for k, v in arrays.items() if k != "background"}, params)
    got = gpu.render(prog, w, h)
    assert np.array_equal(got, want), (name, (w, h), lsb_report(got, want))


@pytest.mark.parametrize("seed", range(16))
def test_sampler_formats_filters_and_wraps(gpu, seed):
    """texture() through video.frag on textures of every format (u8, u16, f16, f32), 1-4 components, both filters and wrap modes,
    odd sizes, moved cameras: bit-exact against the oracle"""
    rng = np.random.default_rng(3000 + seed)
    dtype = [np.uint8, np.uint16, np.float16, np.float32][seed % 4]
    components = int(rng.integers(1, 5))
    tw, th = int(rng.integers(1, 40)), int(rng.integers(1, 30))
    if dtype in (np.uint8, np.uint16):
        data = rng.integers(0, np.iinfo(dtype).max + 1, (th, tw, components)).astype(dtype)
    else:
        data = rng.uniform(-0.5, 1.5, (th, tw, components)).astype(dtype)
    filter, rx, ry = ("linear" if seed % 3 else "nearest"), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    w, h = int(rng.integers(3, 200)), int(rng.integers(3, 120))
    u = O.default_uniforms(w, h, iCameraZoom=float(rng.uniform(0.5, 2.5)), iCameraPosition=(float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), 0.0))
    want = O.render("video", u, {0: O.make_texture(data, filter, rx, ry)}, w, h, threads=4)
    prog, _ = gpu.program("video")
    gpu.set_uniforms(prog, u)
    assert gpu.bind(prog, "iVideo", gpu.texture(data, filter, rx, ry))
    got = gpu.render(prog, w, h)
    assert np.array_equal(got, want), ((dtype.__name__, components, (tw, th), filter, rx, ry), lsb_report(got, want))


@pytest.mark.parametrize("seed", range(10))
def test_resolve_random_sizes(gpu, seed):
    rng = np.random.default_rng(4000 + seed)
    w, h = int(rng.integers(2, 150)), int(rng.integers(2, 90))
    factor = float(rng.choice([1.0, 1.25, 1.5, 2.0, 3.0, 4.0]))
    wr, hr = int(w*factor), int(h*factor)
    subsample = int(rng.integers(1, 5))
    screen = rng.integers(0, 256, (hr, wr, 4), dtype=np.uint8)
    want = O.resolve(screen, w, h, subsample, threads=4)
    got = gpu.resolve(screen, w, h, subsample)
    assert np.array_equal(got, want), ((w, h, factor, subsample), lsb_report(got, want))


# ---- the LDS tile of translated fragments under random maps -----------------------------------------------------------------------
TILE_FUZZ = """
uniform vec4 map = vec4(1.0, 0.0, 0.0, 1.0);
uniform vec2 drift = vec2(0.0);
uniform vec2 spacing = vec2(0.01);
uniform int taps = 2;
uniform float bend = 0.0;
void main() {
    vec2 centre = mat2(map.x, map.y, map.z, map.w)*(astuv - 0.5) + 0.5 + drift + bend*sin(7.0*astuv.yx);
    vec4 sum = vec4(0.0);
    for (int x = -taps; x <= taps; x++)
        for (int y = -taps; y <= taps; y++)
            sum += texture(background, centre + vec2(x, y)*spacing)*(1.0 + 0.1*float(x - y));
    fragColor = sum/float((2*taps + 1)*(2*taps + 1));
}
"""


@pytest.fixture(scope="module")
def tile_programs(gpu):
    """the same fragment translated with and without the tile (two code objects for the whole sweep)"""
    import ctypes as C
    import os
    from pathlib import Path

    from shaderflow_amd import _native as N
    from shaderflow_amd import glsl2hip
    programs = {}
    before = os.environ.get("SHADERFLOW_JIT_TILE")
    for tile in ("1", "0"):
        os.environ["SHADERFLOW_JIT_TILE"] = tile
        translation = glsl2hip.translate(TILE_FUZZ, [("sampler2D", "background")])
        assert (translation.tiled_sampler is not None) == (tile == "1")
        code = glsl2hip.compile(translation, cache=Path(__file__).parent.parent/"build"/"jit")
        names = [b.name.encode() for b in translation.bindings]
        table = (N.Binding*len(names))(*[N.Binding(n, int(b.sampler), b.slot, b.count, int(b.integer)) for n, b in zip(names, translation.bindings)])
        handle = N.Handle()
        N.check(gpu.lib.sfx_program_load(gpu.ctx.handle, code, len(code), table, len(names), C.byref(handle)))
        programs[tile] = handle
    if before is None:
        os.environ.pop("SHADERFLOW_JIT_TILE", None)
    else:
        os.environ["SHADERFLOW_JIT_TILE"] = before
    yield programs
    for handle in programs.values():
        gpu.lib.sfx_program_destroy(handle)


@pytest.mark.parametrize("seed", range(40))
def test_tiled_translated_fragment_random_maps(gpu, tile_programs, seed):
    """Scaled, sheared, mirrored, shifted and bent tap patterns over textures of every format, both filters and wraps, on every kernel of
    the code object (plain, fused 1x / 2x / 4x): with the tile and without it the frames are the same bytes (floats for the plain kernel),
    whether the probed box holds all the taps (affine maps), some (bends, footprints beyond the tile's capacity) or none"""
    rng = np.random.default_rng(4000 + seed)
    kind = ["rgba8", "rgb8", "r32f", "rgba16"][seed % 4]
    tw, th = int(rng.integers(3, 300)), int(rng.integers(2, 200))
    if kind == "rgba8":
        data = rng.integers(0, 256, (th, tw, 4), dtype=np.uint8)
    elif kind == "rgb8":
        data = rng.integers(0, 256, (th, tw, 3), dtype=np.uint8)
    elif kind == "r32f":
        data = rng.random((th, tw, 1), dtype=np.float32)
    else:
        data = rng.integers(0, 65536, (th, tw, 4), dtype=np.uint16)
    filter = "linear" if rng.random() < 0.7 else "nearest"
    repeat = (bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
    texture = gpu.texture(data, filter, *repeat)
    scale = float(rng.choice([0.05, 0.3, 1.0, 1.0, 2.5, -1.0]))
    shear = float(rng.choice([0.0, 0.0, 0.2, -0.6]))
    values = {"map": (scale, shear, -shear*float(rng.random() < 0.5), scale*float(rng.choice([1.0, 0.5, -1.0]))),
              "drift": (float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-1.5, 1.5))),
              "spacing": (float(rng.uniform(0.2, 3.0))/tw, float(rng.uniform(0.2, 3.0))/th),
              "bend": float(rng.choice([0.0, 0.0, 0.02, 0.3]))}
    taps = int(rng.integers(0, 5))
    w, h = int(rng.integers(5, 400)), int(rng.integers(3, 90))
    ssaa, subsample = [(0, 0), (1, 1), (2, 2), (2, 1), (4, 4), (4, 2)][seed % 6]
    frames = []
    for tile in ("1", "0"):
        prog = tile_programs[tile]
        gpu.set_uniforms(prog, O.default_uniforms(w, h, iSSAA=float(max(ssaa, 1))))
        for key, value in values.items():
            assert gpu.set_values(prog, key, value)
        assert gpu.set_values(prog, "taps", taps, integer=True)
        gpu.bind(prog, "background", texture)
        frames.append(gpu.render(prog, w, h, comps=4, dtype=np.float32).view(np.uint32) if ssaa == 0 else gpu.render_resolve(prog, w, h, ssaa, subsample))
    assert np.array_equal(frames[0], frames[1]), (kind, filter, repeat, values, taps, (w, h), (ssaa, subsample))


@pytest.mark.parametrize("seed", range(32))
def test_visualizer_pixel_tier_random_sizes_spectra_and_axis_cameras(gpu, seed):
    """Round 6's pixel tier of the strip kernels (visualizer.frag:36-62's position-only gains once per output pixel, for the wave tiles a
    per-frame classification clears): random frame sizes (odd ones, sizes that leave partial blocks and partial waves), 2x and 4x SSAA,
    spectrogram columns from silence over smooth ones to a single loud bin, loudness / flash / time at random, the identity camera and
    zoomed / panned ones, repeating and clamped backgrounds — whole frames against the oracle within 1 LSB, and the tier really ran."""
    from tests.helpers import smooth_spectrum
    rng = np.random.default_rng(6000 + seed)
    ssaa = 2 if seed % 4 else 4
    w, h = [(int(rng.integers(300, 1400)), int(rng.integers(170, 800))), (1280, 720), (1921, 1079), (960, 540)][seed % 4] if ssaa == 2 else (int(rng.integers(200, 700)), int(rng.integers(120, 400)))
    bg_size = [(384, 216), (1920, 1080), (640, 360)][seed % 3]
    u, arrays, params = visualizer_inputs(w, h, seed=seed, volume=float(rng.choice([0.0, 0.3, 0.8, 1.2])), bg_size=bg_size, time=float(rng.uniform(0, 60)), std=float(rng.uniform(0, 0.8)))
    kind = seed % 3
    if kind == 0:
        arrays["iSpectrogram"] = smooth_spectrum(seed=seed)
    elif kind == 1:
        arrays["iSpectrogram"] = np.zeros((115, 1, 2), np.float32)
    else:
        column = np.full((115, 1, 2), 1.0e-3, np.float32)
        column[int(rng.integers(2, 112)), 0, int(rng.integers(0, 2))] = float(rng.uniform(200.0, 3000.0))
        arrays["iSpectrogram"] = column
    params["background"] = ("linear", bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
    if seed % 2:
        u.iCameraZoom = float(rng.choice([0.7, 0.85, 1.2, 1.5]))
        u.iCameraPosition[0] = float(rng.uniform(-0.15, 0.15)); u.iCameraPosition[1] = float(rng.uniform(-0.1, 0.1))
    u.iSSAA = float(ssaa)
    prog, _ = gpu.program("visualizer")
    gpu.set_uniforms(prog, u)
    gpu_bind_all(gpu, prog, arrays, params)
    gpu.ctx.tile_misses()
    got = gpu.render_resolve(prog, w, h, ssaa, 2)
    per_sample_waves = gpu.ctx.tile_misses()
    kernel = gpu.lib.sfx_last_kernel().decode()
    screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, threads=16)
    want = O.resolve(screen, w, h, 2, threads=16)
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1, (seed, (w, h), ssaa, kernel, lsb_report(got, want), np.argwhere(d > 1)[:5].tolist())
    if kernel.startswith("k_visualizer_strip<"):
        walk = int(kernel.split("<")[1].split(",")[3])
        waves = -(-w*ssaa//64)*(-(-h*ssaa//walk))
        assert 0 < per_sample_waves < waves, (seed, kernel, per_sample_waves, waves)       # both paths populated (the disc's edge is always per sample)
