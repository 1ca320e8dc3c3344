"""
Mipmapped textures on the GPU (texture.py:116-137, 274-283): the chain kernel, the trilinear / nearest-level
sampler with quad derivatives, the Python API's build order — against the oracle's specification arithmetic (what the kernels
restate: ≤ 1 LSB / to rounding) and against the reference's own frames on Mesa llvmpipe (tests/golden/mip.npz), within the bounds
tests/test_oracle_mip.py derives from llvmpipe's measured approximations (its lambda sits up to 0.043 below log2 rho).
"""
from pathlib import Path

import numpy as np
import pytest

from oracle import binding as O
from shaderflow_amd import _native as N
from tests.helpers import Gpu, lsb_report, mip_probe_texture, oracle_textures, visualizer_inputs
from tests.test_oracle_mip import PROBES, probe as oracle_probe

pytestmark = pytest.mark.gpu
G = np.load(Path(__file__).parent/"golden"/"mip.npz")


@pytest.fixture()
def gpu():
    g = Gpu()
    yield g
    g.close()


@pytest.mark.parametrize("shape,dtype", [((6, 8), np.uint8), ((5, 7), np.uint8), ((4, 13), np.uint8), ((6, 8), np.float32), ((270, 480), np.uint8),
                                          ((1080, 1920), np.uint8), ((3, 1), np.float32), ((1, 1), np.uint8), ((37, 64), np.float16), ((9, 5), np.uint16)])
def test_chain_kernel_builds_the_oracles_levels(gpu, shape, dtype):
    """sfx_texture_build_mipmaps level by level against sfo_build_mipmaps (float weights): identical bytes — including the floored
    odd extents, three-component textures (RGB8 backgrounds) and a full-size background; against llvmpipe's own levels within 1 LSB"""
    rng = np.random.default_rng(shape[0]*131 + shape[1])
    components = 3 if shape == (270, 480) else 4
    data = (rng.integers(0, 256 if dtype == np.uint8 else 65536, (*shape, components)).astype(dtype) if np.issubdtype(dtype, np.integer)
            else rng.random((*shape, components)).astype(dtype))
    tag = f"{shape[1]}x{shape[0]}.{np.dtype(dtype).name}"
    if f"levels.{tag}.0" in G.files:
        data = G[f"levels.{tag}.0"]
    handle = gpu.texture(data, "linear", True, True)
    N.check(gpu.lib.sfx_texture_build_mipmaps(handle))
    want = O.build_mipmaps(O.make_texture(data, "linear", True, True))
    for level in range(1, want.levels):
        w, h = max(1, data.shape[1] >> level), max(1, data.shape[0] >> level)
        got = np.empty((h, w, components), dtype)
        N.check(gpu.lib.sfx_texture_read_level(handle, level, got.ctypes.data, got.nbytes))
        expected = O.mip_level(want, level)
        if np.issubdtype(dtype, np.integer):
            assert np.array_equal(got, expected), (tag, level, lsb_report(got, expected))
        else:
            assert np.array_equal(got.view(np.uint16 if dtype == np.float16 else np.uint32), expected.view(np.uint16 if dtype == np.float16 else np.uint32)), (tag, level)
        if f"levels.{tag}.{level}" in G.files and dtype == np.uint8:
            assert np.abs(got.astype(int) - G[f"levels.{tag}.{level}"].astype(int)).max() <= 1
    # an invalid level and an unknown filter are refused
    scratch = np.empty(64, np.uint8)
    assert gpu.lib.sfx_texture_read_level(handle, max(1, want.levels), scratch.ctypes.data, scratch.nbytes) != 0
    assert gpu.lib.sfx_texture_params(handle, 4, 1, 1) != 0


def product_probe(texels, scale, rotation, filter, *, stale, ssaa=1, width=96, height=54):
    """The probe through the Python API as a user writes it: ShaderTexture(mipmaps=True) + a fragment of their own (translated at run
    time). `stale`: from_numpy alone; otherwise followed by repeat(True), which applies — and builds the chain of the data."""
    from shaderflow_amd.scene import ShaderScene
    from shaderflow_amd.texture import ShaderTexture
    c, s = float(np.cos(rotation)), float(np.sin(rotation))
    fragment = (f"void main() {{ vec2 p = astuv*vec2({float(scale[0])!r}, {float(scale[1])!r}); "
                f"fragColor = texture(probe, vec2({c!r}*p.x - {s!r}*p.y, {s!r}*p.x + {c!r}*p.y)); }}")

    class Probe(ShaderScene):
        def build(self):
            texture = ShaderTexture(scene=self, name="probe", filter=filter, mipmaps=True)
            texture.from_numpy(np.flipud(texels))
            if not stale:
                texture.repeat(True)
            self.shader.fragment = fragment

    scene = Probe()
    raw = scene.main(width=width, height=height, ssaa=ssaa, subsample=1 if ssaa == 1 else 2, fps=60.0, time=1/60, output=bytes)
    frame = np.frombuffer(raw, np.uint8).reshape(height, width, 3)
    return frame, scene


@pytest.mark.parametrize("dtype", ["uint8", "float32"])
@pytest.mark.parametrize("tag", ["magnified", "x1.6", "x3.3", "x9", "nearest.x3.3", "x3.3.stale"])
def test_probes_through_the_python_api(dtype, tag):
    """ShaderTexture(mipmaps=True) under a fragment of the user's own, exported by scene.main(): the unfused kernel in its quad
    layout (a mipmapped sampler makes the program a derivative taker: sfx_program_fusable refuses ssaa 1), then final.glsl at
    subsample 1 (a copy). Against the oracle's specification arithmetic ≤ 1 LSB; against the reference's frame within the measured
    distance of llvmpipe's approximations."""
    filter = "nearest" if tag.startswith("nearest") else "linear"
    key = tag.replace("nearest.", "").replace(".stale", "")
    stale = tag.endswith("stale")
    scale, rotation = PROBES[key]
    texels = mip_probe_texture(64, 48, np.dtype(dtype))
    got, scene = product_probe(texels, scale, rotation, filter, stale=stale)
    want = oracle_probe(texels, scale, rotation, filter, stale=stale)[..., :3]
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d == 0).mean() > 0.98, (tag, lsb_report(got, want))
    reference = G[f"probe.{dtype}.{filter}.{key}{'.stale' if stale else ''}"][..., :3]
    d = np.abs(got.astype(int) - reference.astype(int))
    limit = 1 if key == "magnified" else (12 if stale else (6 if key == "x9" else 4))
    assert d.max() <= limit and (d <= 2).mean() > 0.9, (tag, np.bincount(d.ravel())[:14])


@pytest.mark.parametrize("tag", ["x3.3", "nearest.x3.3", "x9"])
def test_tiled_translation_under_a_mipmapped_sampler_takes_the_quad_layout(monkeypatch, tag):
    """a translated fragment that got an LDS tile walks four rows per lane (JitShader<F, false, true>::ROWS_1X), so its
    lanes cannot form 2 x 2 quads — with a mipmapped sampler bound the differences across a "quad" would be between unrelated pixels
    and the level of detail silently wrong. The library launches the code object's untiled twin (sfx_jit_render_quads) for that draw.
    SHADERFLOW_JIT_TILE=probe forces the tile for the probe's sampler: same frames as the untiled translation pinned above — the
    oracle's specification arithmetic ≤ 1 LSB, the reference's llvmpipe frame within its measured distance"""
    monkeypatch.setenv("SHADERFLOW_JIT_TILE", "probe")
    filter = "nearest" if tag.startswith("nearest") else "linear"
    key = tag.replace("nearest.", "")
    scale, rotation = PROBES[key]
    texels = mip_probe_texture(64, 48, np.uint8)
    got, scene = product_probe(texels, scale, rotation, filter, stale=False)
    assert "#define SF_JIT_TILE_SLOT" in scene.shader._translation_text            # the translation under test really is the tiled one
    want = oracle_probe(texels, scale, rotation, filter, stale=False)[..., :3]
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d == 0).mean() > 0.98, (tag, lsb_report(got, want))
    reference = G[f"probe.uint8.{filter}.{key}"][..., :3]
    d = np.abs(got.astype(int) - reference.astype(int))
    assert d.max() <= (6 if key == "x9" else 4) and (d <= 2).mean() > 0.9, (tag, np.bincount(d.ravel())[:14])
    monkeypatch.setenv("SHADERFLOW_JIT_TILE", "0")
    untiled, _ = product_probe(texels, scale, rotation, filter, stale=False)
    assert np.array_equal(got, untiled)


def test_fused_kernel_at_two_supersamples_takes_the_derivatives_across_the_pixels_quad():
    """ssaa 2: the four supersamples of a pixel are the four lanes of a quad in the fused kernel, so a mipmapped sampler stays fusable
    there (sfx_program_fusable) — level of detail in SUPERSAMPLE units, as the reference's render at render_resolution has it"""
    texels = mip_probe_texture(64, 48, np.uint8)
    scale, rotation = PROBES["x3.3"]
    got, scene = product_probe(texels, scale, rotation, "linear", stale=False, ssaa=2)
    assert N.lib().sfx_program_fusable(scene.shader.program, 2) == 1 and N.lib().sfx_program_fusable(scene.shader.program, 1) == 0
    screen = oracle_probe(texels, scale, rotation, "linear", stale=False, width=192, height=108)
    want = O.resolve(screen, 96, 54, 2)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1, lsb_report(got, want)


def test_builtin_fragment_over_a_mipmapped_background(gpu):
    """visualizer.frag with a mipmapped 480x270 background under 160x90 (2.8 texels per pixel): the fast kernels step aside (their
    tables and tiles hold level 0), the plain kernel runs in the quad layout; against the oracle's three-evaluation scheme ≤ 1 LSB,
    and within llvmpipe's measured distance of the reference's frame"""
    u, arrays, params = visualizer_inputs(160, 90, seed=21, volume=0.5, bg_size=(480, 270))
    prog, fallback = gpu.program("visualizer")
    assert not fallback
    gpu.set_uniforms(prog, u)
    for name, data in arrays.items():
        handle = gpu.texture(data, *params[name])
        if name == "background":
            N.check(gpu.lib.sfx_texture_build_mipmaps(handle))
            N.check(gpu.lib.sfx_texture_params(handle, 2, 1, 1))                     # SFX_LINEAR_MIPMAP_LINEAR
        assert gpu.bind(prog, name, handle)
    got = gpu.render(prog, 160, 90)
    assert "k_render<" in N.lib().sfx_last_kernel().decode() and "Plain" in N.lib().sfx_last_kernel().decode()
    textures = oracle_textures(arrays, params)
    O.build_mipmaps(textures["background"])
    want = O.render("visualizer", u, textures, 160, 90, threads=8)
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d == 0).mean() > 0.95, lsb_report(got, want)
    reference = G["visualizer.mip.image"]
    d = np.abs(got.astype(int) - reference.astype(int))
    assert d.max() <= 6 and (d <= 1).mean() > 0.93 and (d <= 2).mean() > 0.99, np.bincount(d.ravel())[:8]
    # the same program at 2x SSAA through the fused kernel (quads = the pixel's supersamples): oracle at render resolution, resolved
    u.iSSAA = 2.0
    gpu.set_uniforms(prog, u)
    fused = gpu.render_resolve(prog, 160, 90, 2, 2)
    screen = O.render("visualizer", u, textures, 320, 180, threads=8)
    assert np.abs(fused.astype(int) - O.resolve(screen, 160, 90, 2).astype(int)).max() <= 1
