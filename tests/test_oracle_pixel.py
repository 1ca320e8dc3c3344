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
