"""
Mipmapped textures (texture.py:116-137, 274-283 — `mipmaps=True` → build_mipmaps() + LINEAR_MIPMAP_LINEAR): the
oracle against tests/golden/mip.npz, rendered by the reference itself on Mesa llvmpipe (make_golden_mip.py). CPU only.

Two oracles in one, as for the bilinear filter (test_oracle_mesa.py): the SPECIFICATION's arithmetic — float weights, lambda =
log2(rho), what the HIP kernels compute (tests/test_gpu_mip.py compares them with it) — and, under `O.llvmpipe_filter()`, llvmpipe's
own — fixed-point filter, lambda = log2(rho²)/2 with log2 read off the float's exponent and mantissa, an 8-bit blend between levels —
which must reproduce the goldens (unorm8: bit for bit). The first is then held to the goldens within what the second explains.
"""
from pathlib import Path

import numpy as np
import pytest

from oracle import binding as O
from tests.helpers import mip_probe_texture, oracle_textures, visualizer_inputs

G = np.load(Path(__file__).parent/"golden"/"mip.npz")
PROBES = {"magnified": ((0.6, 0.5), 0.2), "x1.6": ((2.4, 2.0), 0.15), "x3.3": ((5.0, 4.1), -0.4), "x9": ((13.0, 11.0), 0.0)}


@pytest.mark.parametrize("tag", ["8x6.uint8", "7x5.uint8", "13x4.uint8", "8x6.float32", "7x5.float32"])
def test_mip_chain_is_the_half_size_bilinear_image_of_the_level_above(tag):
    """glGenerateMipmap on the implementation behind the goldens: every level from the one above — the 2x2 box mean for even
    extents, two-texel taps for floored odd ones. unorm8: llvmpipe's levels bit for bit under the switch, within 1 LSB of the
    float-weight chain (what the kernels build); float32: to rounding."""
    level0 = G[f"levels.{tag}.0"]
    for llvmpipe in (False, True):
        texture = O.make_texture(level0, "linear", True, True)
        if llvmpipe:
            with O.llvmpipe_filter():
                O.build_mipmaps(texture)
        else:
            O.build_mipmaps(texture)
        assert texture.levels == O.lib().sfo_mip_levels(level0.shape[1], level0.shape[0]) and texture.filter == 2
        for k in range(1, texture.levels):
            want, got = G[f"levels.{tag}.{k}"], O.mip_level(texture, k)
            assert got.shape == want.shape, (tag, k, got.shape, want.shape)
            if level0.dtype == np.uint8:
                d = np.abs(got.astype(int) - want.astype(int))
                assert d.max() <= (0 if llvmpipe else 1), (tag, k, llvmpipe, d.max())
            else:
                assert np.abs(got - want).max() < 4e-7, (tag, k)


def test_level_of_detail_the_reference_implementation_selects():
    """lod.*: llvmpipe's lambda, measured with constant levels. It is log2(rho) taken as HALF the exponent-plus-mantissa reading of
    rho² — exact at powers of two, up to 0.043 below in between, the same for a rotated footprint (the Euclidean rho of the
    specification, not a max-norm). The specification's own log2(rho) is what the kernels use; the difference bounds what a
    golden may deviate by: 0.043 x the difference between two neighbouring levels."""
    rho, measured = G["lod.rho"], G["lod.lambda"]
    exact = np.maximum(0.0, np.log2(rho))
    rho2 = (rho.astype(np.float32)**2).astype(np.float32)
    mantissa, exponent = np.frexp(rho2)
    model = np.where(rho <= 1.0, 0.0, 0.5*((exponent - 1) + (2.0*mantissa - 1.0)))
    assert np.abs(model - measured).max() < 2e-3, np.abs(model - measured).max()
    assert (exact - measured).min() > -2e-3 and (exact - measured).max() < 0.0432
    for rotation, r, low, high in G["lod.rotated"]:
        assert abs(low - high) < 2e-3 and abs(low - 0.5*np.log2(r*r)) < 0.0432           # rotation changes nothing: Euclidean rho
        assert abs(low - np.log2(r*max(abs(np.cos(rotation)), abs(np.sin(rotation))))) > 0.04 or rotation == 0
    # the blend of two unorm8 levels (k·16 and (k+1)·16): weight trunc(frac·256), a + ((w·(b − a) + 128) >> 8)
    for r, value in G["lod.blend_u8"]:
        r32 = np.float32(r)*np.float32(r)
        m, e = np.frexp(r32)
        lam = 0.0 if r <= 1.0 else 0.5*((e - 1) + (2.0*float(m) - 1.0))
        w = int((lam - np.floor(lam))*256.0)
        base = int(np.floor(lam))*16
        assert round(value) == base + ((w*16 + 128) >> 8), (r, value)


def probe(texels: np.ndarray, scale, rotation: float, filter: str, *, stale: bool, width=96, height=54) -> np.ndarray:
    """texture(probe, R·(astuv·S)) on the oracle: the `sampler` test fragment through sfo_render would need a new fragment id; the
    quad derivatives of an AFFINE coordinate are the same everywhere, so the level of detail is set up once and O.sample does the rest"""
    texture = O.make_texture(texels, filter, True, True)
    O.build_mipmaps(texture, source=np.zeros_like(texels) if stale else None)
    c, s = np.float32(np.cos(rotation)), np.float32(np.sin(rotation))
    out = np.zeros((height, width, 4), np.uint8)
    one, half = np.float32(1), np.float32(0.5)

    def coordinate(i, j):
        px = (np.float32(i) + half)/np.float32(width)*np.float32(scale[0])
        py = (np.float32(j) + half)/np.float32(height)*np.float32(scale[1])
        return c*px - s*py, s*px + c*py
    for j in range(height):
        for i in range(width):
            out[j, i] = np.rint(np.clip(O.sample_quad(texture, coordinate(i, j), coordinate(i ^ 1, j), coordinate(i, j ^ 1)), 0, 1)*255)
    return out


@pytest.mark.parametrize("dtype", ["uint8", "float32"])
@pytest.mark.parametrize("tag", ["magnified", "x1.6", "x3.3", "x9", "nearest.x3.3", "x3.3.stale"])
def test_mipmapped_probes_rendered_by_the_reference(dtype, tag):
    """probe.*: the repository's probe fragment as scene.shader.fragment of a scene of the reference, `ShaderTexture(mipmaps=True)`.
    `stale`: from_numpy() alone — make() → apply() builds the chain BEFORE write() fills level 0 (texture.py:330-335), so the
    minified image blends the data with an EMPTY chain; the product mirrors that order (shaderflow_amd/texture.py)."""
    filter = "nearest" if tag.startswith("nearest") else "linear"
    key = tag.replace("nearest.", "").replace(".stale", "")
    scale, rotation = PROBES[key]
    texels = mip_probe_texture(64, 48, np.dtype(dtype))
    want = G[f"probe.{dtype}.{filter}.{key}{'.stale' if tag.endswith('stale') else ''}"]
    with O.llvmpipe_filter():
        model = probe(texels, scale, rotation, filter, stale=tag.endswith("stale"))
    d = np.abs(model.astype(int) - want.astype(int))
    if dtype == "uint8":
        assert d.max() <= 1 and (d == 0).mean() > 0.97, (tag, np.bincount(d.ravel())[:4])     # llvmpipe's arithmetic (8-bit stages; coordinates as interpolated varyings)
    else:
        assert d.max() <= 1, (tag, np.bincount(d.ravel())[:4])
    spec = probe(texels, scale, rotation, filter, stale=tag.endswith("stale"))
    d = np.abs(spec.astype(int) - want.astype(int))
    # the specification's lambda is up to 0.043 above llvmpipe's: 0.043 x (level difference) + the 8-bit stages
    # (deeper levels add the chain's own roundings: llvmpipe rounds every unorm8 level through its fixed-point filter)
    limit = 1 if key == "magnified" else (12 if tag.endswith("stale") else (6 if key == "x9" else 4))
    within = 0.5 if (tag.endswith("stale") or key == "x9") else 0.85
    assert d.max() <= limit and (d <= 1).mean() > within and (d <= 2).mean() > 0.9, (tag, np.bincount(d.ravel())[:14])


def test_reference_fragment_over_a_mipmapped_background():
    """visualizer.frag with `background.mipmaps = True` (480x270 under 160x90: 2.8 texels per pixel, lambda ≈ 1.4): every tap of the
    blur blends levels 1 and 2 — through sfo_render's three-evaluation quad scheme"""
    u, arrays, params = visualizer_inputs(160, 90, seed=21, volume=0.5, bg_size=(480, 270))
    want = G["visualizer.mip.image"]
    for llvmpipe in (True, False):
        textures = oracle_textures(arrays, params)
        if llvmpipe:
            with O.llvmpipe_filter():
                O.build_mipmaps(textures["background"])
                got = O.render("visualizer", u, textures, 160, 90, threads=8)
        else:
            O.build_mipmaps(textures["background"])
            got = O.render("visualizer", u, textures, 160, 90, threads=8)
        d = np.abs(got.astype(int) - want.astype(int))
        # a pure-noise background: neighbouring levels differ by tens of LSB, and the specification's lambda sits up to 0.043 above llvmpipe's
        assert d.max() <= (1 if llvmpipe else 6), (llvmpipe, np.bincount(d.ravel())[:6])
        assert (d <= 1).mean() > (0.999 if llvmpipe else 0.93) and (d <= 2).mean() > 0.99, (llvmpipe, np.bincount(d.ravel())[:6])
