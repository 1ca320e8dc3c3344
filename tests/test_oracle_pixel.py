"""
Pixel half of the oracle (oracle/sfo_pixel.c, oracle/sfo_math.h): the reference holds no golden
images, so these are the closed-form / known-answer checks SURVEY.md §8c lists.
CPU only.
"""
import numpy as np
import pytest

from oracle import binding as O


def ulp_error(got32, want64):
    want32 = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    ulp[ulp == 0] = np.finfo(np.float32).tiny
    return np.abs(got32.astype(np.float64) - want64)/ulp


def test_sfmath_accuracy():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-200, 200, 3000), rng.uniform(-7, 7, 2000), [0.0, 1e-8, -1e-8, 3.1415927, 6.2831855]]).astype(np.float32)
    assert ulp_error(O.math("sin", x), np.sin(x.astype(np.float64)))[np.abs(np.sin(x.astype(np.float64))) > 1e-4].max() <= 4
    assert ulp_error(O.math("cos", x), np.cos(x.astype(np.float64)))[np.abs(np.cos(x.astype(np.float64))) > 1e-4].max() <= 4
    assert np.abs(O.math("sin", x) - np.sin(x.astype(np.float64))).max() < 2.5e-7
    y = rng.uniform(-3, 3, 4000).astype(np.float32); z = rng.uniform(-3, 3, 4000).astype(np.float32)
    assert ulp_error(O.math("atan2", y, z), np.arctan2(y.astype(np.float64), z.astype(np.float64))).max() <= 4
    p = np.exp(rng.uniform(-30, 30, 4000)).astype(np.float32)
    want = np.log2(p.astype(np.float64))
    assert np.abs(O.math("log2", p) - want).max() <= 4*np.spacing(np.float32(1.0))*np.maximum(1.0, np.abs(want)).max()
    e = rng.uniform(-20, 20, 4000).astype(np.float32)
    assert ulp_error(O.math("exp2", e), np.exp2(e.astype(np.float64))).max() <= 4
    b = rng.uniform(0.0, 3.0, 4000).astype(np.float32); ex = rng.uniform(0.05, 6.0, 4000).astype(np.float32)
    want = np.power(b.astype(np.float64), ex.astype(np.float64))
    assert np.allclose(O.math("pow", b, ex), want, rtol=3e-6, atol=1e-30)
    # exact identities that the fragments rely on
    assert O.math("pow", np.float32(0.0), np.float32(2.5)) == 0.0
    assert O.math("pow", np.float32(1.0), np.float32(2.5)) == 1.0
    assert O.math("sin", np.float32(0.0)) == 0.0 and O.math("cos", np.float32(0.0)) == 1.0
    assert np.isnan(O.math("pow", np.float32(-1.0), np.float32(0.5)))


def test_sampler_nearest_ramp():
    ramp = np.arange(8, dtype=np.float32).reshape(8, 1, 1).repeat(2, axis=2)      # (h=8, w=1, RG)
    ramp[:, :, 1] += 100
    tex = O.make_texture(ramp, "nearest", True, False)
    for j in range(8):
        c = O.sample(tex, 0.0, (j + 0.5)/8)
        assert tuple(c) == (j, 100 + j, 0.0, 1.0)
    assert O.sample(tex, 0.0, -0.3)[0] == 0 and O.sample(tex, 0.0, 1.7)[0] == 7      # clamp in y
    assert O.sample(tex, 5.25, 0.5)[0] == 4                                         # repeat in x


def test_sampler_bilinear_wrap_and_unorm():
    img = np.array([[[0, 255, 10]], [[255, 0, 30]]], np.uint8).transpose(1, 0, 2).copy()   # (h=1, w=2, RGB)
    rep = O.make_texture(img, "linear", True, True)
    cl = O.make_texture(img, "linear", False, False)
    assert np.allclose(O.sample(rep, 0.25, 0.5), [0, 1, 10/255, 1], atol=1e-7)
    assert np.allclose(O.sample(rep, 0.5, 0.5), [0.5, 0.5, 20/255, 1], atol=1e-7)
    assert np.allclose(O.sample(rep, 0.0, 0.5), [0.5, 0.5, 20/255, 1], atol=1e-7)   # wraps to texel 1
    assert np.allclose(O.sample(cl, 0.0, 0.5), [0, 1, 10/255, 1], atol=1e-7)        # clamps to texel 0
    assert np.allclose(O.sample(cl, 1.0, 0.5), [1, 0, 30/255, 1], atol=1e-7)


def test_multishader_closed_form():
    """demo.py:67-89: rgb = (stuv.x, 1 - stuv.x, 0) after adding the child pass"""
    w, h = 64, 36
    u = O.default_uniforms(w, h)
    child = O.render("multi_child", u, {}, w, h)
    tex = O.make_texture(child, "linear", True, True)
    main = O.render("multi_main", u, {"child": tex}, w, h)
    aspect = np.float32(w)/np.float32(h)
    x = (np.arange(w, dtype=np.float32) + 0.5)/w
    stuv = ((2*x - 1)*aspect + 1)/2
    want_r = np.rint(np.clip(stuv, 0, 1)*255)
    want_g = np.rint(np.clip(1 - stuv, 0, 1)*255)
    assert np.abs(main[:, :, 0].astype(int) - want_r[None, :]).max() <= 1
    assert np.abs(main[:, :, 1].astype(int) - want_g[None, :]).max() <= 1
    assert (main[:, :, 2] == 0).all() and (main[:, :, 3] == 255).all()


def test_shadertoy_closed_form():
    w, h = 48, 27
    u = O.default_uniforms(w, h, iTime=1.25)
    img = O.render("shadertoy", u, {}, w, h)
    aspect = w/h
    x = ((2*(np.arange(w) + 0.5)/w - 1)*aspect + 1)/2
    y = ((2*(np.arange(h) + 0.5)/h - 1) + 1)/2
    r = 0.5 + 0.5*np.cos(1.25 + x)[None, :].repeat(h, 0)
    g = 0.5 + 0.5*np.cos(1.25 + y + 2)[:, None].repeat(w, 1)
    assert np.abs(img[:, :, 0].astype(int) - np.rint(np.clip(r, 0, 1)*255)).max() <= 1
    assert np.abs(img[:, :, 1].astype(int) - np.rint(np.clip(g, 0, 1)*255)).max() <= 1


@pytest.mark.parametrize("ssaa,subsample", [(2, 2), (4, 2), (4, 4), (1, 1), (3, 3)])
def test_resolve_box_mean(ssaa, subsample):
    """final.glsl: taps hit texel centres (s == k) or 2x2 centres (s == 2k) → exact box mean"""
    rng = np.random.default_rng(1)
    w, h = 20, 12
    screen = rng.integers(0, 256, (h*ssaa, w*ssaa, 4), dtype=np.uint8)
    out = O.resolve(screen, w, h, subsample)
    box = screen[:, :, :3].astype(np.float64).reshape(h, ssaa, w, ssaa, 3).mean(axis=(1, 3))
    assert np.abs(out.astype(np.float64) - box).max() <= 0.5 + 1e-3


def test_resolve_tent_when_not_supersampled():
    """s = 1, k = 2: taps at ±1/4 px → 3x3-footprint tent"""
    w, h = 16, 10
    screen = np.zeros((h, w, 4), np.uint8)
    screen[5, 7] = 255
    out = O.resolve(screen, w, h, 2)[:, :, 0].astype(float)/255
    k = np.array([0.25, 1.5, 0.25])/2             # per-axis weights: mean of (.75,.25) and (.25,.75) taps
    want = np.outer(k, k)
    assert np.allclose(out[4:7, 6:9], want, atol=1/255)
    assert out.sum() == pytest.approx(1.0, abs=0.03)


def test_default_scene_properties():
    """fragment/default.glsl at 256x256 (BASELINE config 1): symmetric ring, opaque, vignette"""
    w = h = 128
    u = O.default_uniforms(w, h, iTau=0.0)
    img = O.render("default", u, {}, w, h)
    assert (img[:, :, 3] == 255).all()
    centre = img[h//2, w//2, :3]
    assert np.abs(centre.astype(int) - round(0.18*255)).max() <= 2            # inside the circle: 0.18 + ring
    corner = img[0, 0, :3]
    assert corner.max() < centre.min()                                        # vignette darkens corners
    ring = img[h//2, int(w*(0.5 + 0.75/2)) , :3]                              # |uv| = 0.75 = 1/1.333
    assert ring.max() == 255
    # band rendering equals the full frame on those rows
    band = O.render("default", u, {}, w, h, rows=(10, 20), threads=3)
    assert np.array_equal(band[10:20], img[10:20]) and (band[:10] == 0).all()


def test_visualizer_silent_is_background_scaled():
    """visualizer.frag with iAudioVolume = 0: every blur tap equals the centre tap"""
    rng = np.random.default_rng(2)
    w, h = 64, 36
    bg = rng.integers(0, 256, (27, 48, 3), dtype=np.uint8)
    spec = np.zeros((115, 1, 2), np.float32); wave = np.zeros((1, 180, 2), np.float32)
    tex = {
        "background": O.make_texture(bg, "linear", True, True),
        "iSpectrogram": O.make_texture(spec, "nearest", True, False),
        "iWaveform": O.make_texture(wave, "linear", False, False),
    }
    u = O.default_uniforms(w, h, iTime=0.5, iSpectrogramBins=115, iSpectrogramLength=1)
    img = O.render("visualizer", u, tex, w, h, threads=2)
    assert (img[:, :, 3] == 255).all()
    assert img[:, :, :3].std() > 5                                            # background shows through
    # threads do not change results
    assert np.array_equal(img, O.render("visualizer", u, tex, w, h, threads=1))


# ---- multipass / temporal / remaining fragments: known answers ---------------------------------

def life_step(state: np.ndarray, frame: int = 0, period: int = 1) -> np.ndarray:
    h, w = state.shape
    u = O.default_uniforms(w, h, iFrame=frame)
    u.user[0], u.user[1], u.user[2] = w, h, period
    tex = O.make_texture(state.astype(np.float32)[:, :, None], "nearest", True, True)
    return O.render_to("life_simulation", u, {1: tex}, w, h, 1, np.float32)[..., 0]


def test_life_simulation_known_patterns():
    """simulation.glsl:21-53 — still lifes stay, the blinker has period 2, the glider moves one cell diagonally every
    4 generations; cells outside the texture are dead (texelFetch out of range reads zero)"""
    block = np.zeros((8, 8)); block[3:5, 3:5] = 1
    assert np.array_equal(life_step(block), block)
    blinker = np.zeros((7, 7)); blinker[3, 2:5] = 1
    once = life_step(blinker)
    assert np.array_equal(once, blinker.T) and np.array_equal(life_step(once), blinker)
    glider = np.zeros((12, 12)); glider[1, 2] = glider[2, 3] = glider[3, 1] = glider[3, 2] = glider[3, 3] = 1
    state = glider
    for _ in range(4):
        state = life_step(state)
    assert np.array_equal(state, np.roll(np.roll(glider, 1, 0), 1, 1))
    edge = np.zeros((6, 6)); edge[0, 0:3] = 1                        # a blinker on the border: no wrap-around neighbours
    assert np.array_equal(life_step(edge), np.array([[0, 1, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0]] + [[0]*6]*4))
    assert np.array_equal(life_step(blinker, frame=5, period=6), blinker)      # held between life periods (:26-30)
    assert np.array_equal(life_step(blinker, frame=12, period=6), blinker.T)


def test_motionblur_constant_history_closed_form():
    """motionblur.frag:9-14 with every past frame equal to c: 2*c*sum(smoothstep(1, 0, i/T))/T"""
    w, h, T = 16, 9, 10
    frame = np.full((h, w, 4), 100, np.uint8)
    u = O.default_uniforms(w, h, iLayer=1)
    u.user[0] = T
    got = O.render("motionblur", u, {t: O.make_texture(frame, "linear", False, False) for t in range(T)}, w, h)
    x = 1.0 - np.arange(T)/T                                         # smoothstep(1, 0, s) = S(1 - s)
    factors = x*x*(3 - 2*x)
    want = np.rint(255*min(1.0, 2*(100/255)*factors.sum()/T))
    assert np.abs(got[..., :3].astype(int) - want).max() <= 1 and (got[..., 3] == 255).all()
    # an empty history is black: what the first temporal-1 frames of the scene show
    zero = np.zeros((h, w, 4), np.uint8)
    assert not O.render("motionblur", u, {t: O.make_texture(zero) for t in range(T)}, w, h)[..., :3].any()


def test_multipass_layers():
    """multipass.frag:28-45: layer 0 shows the background, layer 1 inverts red left of the centre and blurs the right"""
    w, h = 64, 36
    rng = np.random.default_rng(2)
    background = rng.integers(0, 256, (18, 32, 3), dtype=np.uint8)
    u = O.default_uniforms(w, h, iLayer=0)
    layer0 = O.render("multipass", u, {"background": O.make_texture(background)}, w, h)
    dyn = O.render("dynamics", u, {"background": O.make_texture(background)}, w, h)      # zoom(stuv, 0.85 + 0) ≠ identity, so only the sampler is shared
    assert layer0.shape == dyn.shape and (layer0[..., 3] == 255).all() and layer0[..., :3].std() > 10
    flat = np.zeros((h, w, 4), np.uint8); flat[..., 0] = 40; flat[..., 1] = 90; flat[..., 2] = 200; flat[..., 3] = 255
    u.iLayer = 1
    layer1 = O.render("multipass", u, {0: O.make_texture(flat, "linear", False, False)}, w, h)
    assert (layer1[:, :w//2, 0] == 215).all() and (layer1[:, :w//2, 1] == 90).all() and (layer1[:, :w//2, 2] == 200).all()
    assert np.abs(layer1[:, w//2:, :3].astype(int) - np.array([40, 90, 200])).max() <= 1  # the blur of a constant is the constant


def test_mandelbrot_membership_and_raymarch_depth():
    w, h = 96, 54
    u = O.default_uniforms(w, h)
    img = O.render("mandelbrot", u, {}, w, h, threads=4)
    # c = gluv - (0.5, 0): the pixel whose gluv ≈ (0.5, 0) is c ≈ 0, inside the set → t = pow(0, 20) = 0 → first palette colour
    inside = img[h//2, int((0.5/(16/9) + 1)/2*w)]
    assert np.abs(inside[:3].astype(int) - np.rint(255*np.array([0.01060815, 0.01808215, 0.10018654]))).max() <= 1
    # escape-time image against a float64 restatement (the last palette segment extrapolates: (t - 0.5)*4 reaches 2,
    # shaderflow.glsl:216 — kept as is), compared where the iteration count is far from a float32/float64 disagreement
    ys, xs = np.mgrid[0:h, 0:w]
    c = ((2*(xs + 0.5)/w - 1)*(16/9) - 0.5) + 1j*(2*(ys + 0.5)/h - 1)
    z, count = c.copy(), np.zeros(c.shape, int)
    alive = np.ones(c.shape, bool)
    for _ in range(500):
        alive &= np.abs(z) <= 3.0
        count += alive
        z = np.where(alive, z*z + c, z)
    t = (1 - count/500)**20
    magma = np.array([[0.01060815, 0.01808215, 0.10018654], [0.38092887, 0.12061482, 0.32506528],
                      [0.79650140, 0.10506637, 0.31063031], [0.95922872, 0.53307513, 0.37488950]])
    seg = np.where(t < 0.25, 0, np.where(t < 0.5, 1, 2))
    k = ((t - 0.25*seg)*4)[..., None]
    want = np.rint(255*np.clip(magma[seg]*(1 - k) + magma[seg + 1]*k, 0, 1))
    stable = (count < 30) | (count == 500)
    assert stable.mean() > 0.8
    assert np.abs(img[..., :3].astype(int) - want)[stable].max() <= 2

    ray = O.render("raymarch", u, {}, w, h, threads=4)
    # the central ray meets the first box (centre z = 2, side 1) head-on: one step of 1.5, then a zero step → steps = 1
    assert abs(int(ray[h//2, w//2, 0]) - round(255*0.9)) <= 1 and (ray[..., 3] == 255).all()
    assert ray[0, 0, 0] < ray[h//2, w//2, 0]                          # rays that graze or miss take more steps


def test_tetration_and_video_are_well_formed():
    w, h = 64, 36
    u = O.default_uniforms(w, h)
    img = O.render("tetration", u, {}, w, h, threads=4)
    # k = it / MAX_STEPS is an INTEGER division (tetration.frag:49): value is 0 unless the loop ran to the end → black or a pure hue
    rgb = img[..., :3].astype(int)
    assert set(np.unique(rgb.max(axis=2))) <= {0, 255} and (img[..., 3] == 255).all() and 0 < (rgb.max(axis=2) == 255).mean() < 1
    frame = np.random.default_rng(0).integers(0, 256, (18, 32, 3), dtype=np.uint8)
    vid = O.render("video", u, {0: O.make_texture(frame)}, w, h)
    bg = O.render("multipass", u, {"background": O.make_texture(frame)}, w, h)         # layer 0: stexture(background, stuv)
    assert np.array_equal(vid, bg)                                   # identity camera: iCamera.stuv == stuv


def test_log_matches_libm():
    x = np.exp(np.random.default_rng(3).uniform(-20, 20, 2000)).astype(np.float32)
    assert np.allclose(O.math("log", x), np.log(x.astype(np.float64)), rtol=0, atol=4e-6*20)


def test_float16_textures_and_targets_follow_numpy():
    """numpy float16 is a texture format of the reference (texture.py:28-38 "f2"): reads widen exactly, writes round to nearest even"""
    values = np.arange(65536, dtype=np.uint16).view(np.float16).reshape(256, 256, 1)
    finite = np.isfinite(values[..., 0])
    tex = O.make_texture(values, "nearest", False, False)
    got = np.array([[O.sample(tex, (i + 0.5)/256, (j + 0.5)/256)[0] for i in range(0, 256, 5)] for j in range(0, 256, 3)], np.float32)
    want = values[0:256:3, 0:256:5, 0].astype(np.float32)
    keep = finite[0:256:3, 0:256:5]
    assert np.array_equal(got[keep], want[keep]) and np.isnan(got[~keep & np.isnan(want)]).all()
    u = O.default_uniforms(96, 54, iTime=1.3)
    for fragment in ("shadertoy", "default"):
        wide = O.render_to(fragment, u, {}, 96, 54, 4, np.float32)
        half = O.render_to(fragment, u, {}, 96, 54, 4, np.float16)
        assert np.array_equal(half, wide.astype(np.float16))
    rng = np.random.default_rng(0)
    magnitudes = (np.exp(rng.uniform(-30, 12, 20000))*rng.choice([-1, 1], 20000)).astype(np.float32)   # normals, subnormals, overflow
    probe = O.make_texture(magnitudes.reshape(100, 200, 1), "nearest", False, False)
    u = O.default_uniforms(200, 100)
    rendered = O.render_to("video", u, {0: probe}, 200, 100, 1, np.float16)[..., 0]                      # video.frag copies the texel (identity camera)
    with np.errstate(over="ignore"):
        assert np.array_equal(rendered, magnitudes.reshape(100, 200).astype(np.float16))
