"""Translated blur fragment (17 x 17 bilinear taps of a 1080p RGBA8 texture), fused 2x SSAA at 1080p: LDS tile on / off.
Run on the GPU box: python3 tools/bench_jit_tile.py"""
import ctypes as C, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from oracle import binding as O
from shaderflow_amd import _native as N, glsl2hip
from tests.helpers import Gpu

BLUR = """
uniform float radius = 6.0;
void main() {
    vec4 sum = vec4(0.0);
    float total = 0.0;
    vec2 texel = 1.0/vec2(textureSize(background, 0));
    for (int x = -TAPS; x <= TAPS; x++) {
        for (int y = -TAPS; y <= TAPS; y++) {
            float weight = exp(-float(x*x + y*y)/(radius*radius));
            sum += weight*texture(background, astuv + vec2(x, y)*texel*1.3);
            total += weight;
        }
    }
    fragColor = sum/total;
}
"""
import argparse
parser = argparse.ArgumentParser(description=__doc__)
parser.add_argument("--only", help="one case for a profiler: TAPS:WIDTHxHEIGHT:SSAA:TILE, e.g. 4:3840x2160:2:1")
options = parser.parse_args()
gpu = Gpu()
rng = np.random.default_rng(0)
background = rng.integers(0, 256, (1080, 1920, 4), dtype=np.uint8)
tap_sweep, size_sweep, tile_sweep = (2, 4, 8), ((1920, 1080, 2), (1920, 1080, 1), (3840, 2160, 2)), ("0", "1")
if options.only:
    taps_, size_, ssaa_, tile_ = options.only.split(":")
    tap_sweep, size_sweep, tile_sweep = (int(taps_),), ((*map(int, size_.split("x")), int(ssaa_)),), (tile_,)
for taps in tap_sweep:
    for (w, h, ssaa) in size_sweep:
        line = f"{(2*taps + 1)**2:4d} taps {w}x{h} ssaa {ssaa}:"
        frames = {}
        for tile in tile_sweep:
            os.environ["SHADERFLOW_JIT_TILE"] = tile
            t = glsl2hip.translate(BLUR.replace("TAPS", str(taps)), [("sampler2D", "background")])
            code = glsl2hip.compile(t)
            names = [b.name.encode() for b in t.bindings]
            table = (N.Binding*len(names))(*[N.Binding(n, int(b.sampler), b.slot, b.count, int(b.integer)) for n, b in zip(names, t.bindings)])
            prog = N.Handle()
            N.check(gpu.lib.sfx_program_load(gpu.ctx.handle, code, len(code), table, len(names), C.byref(prog)))
            gpu.bind(prog, "background", gpu.texture(background, "linear", True, True))
            gpu.set_uniforms(prog, O.default_uniforms(w, h, iSSAA=float(ssaa)))
            assert gpu.set_values(prog, "radius", 6.0)             # (initialisers are applied by ShaderProgram, not by sfx_program_load)
            dst = gpu.empty(w, h, 3)
            for _ in range(2):
                N.check(gpu.lib.sfx_render_resolve(prog, dst, ssaa, ssaa))
            N.check(gpu.lib.sfx_ctx_synchronize(gpu.ctx.handle))
            n = 10
            t0 = time.perf_counter()
            for _ in range(n):
                N.check(gpu.lib.sfx_render_resolve(prog, dst, ssaa, ssaa))
            N.check(gpu.lib.sfx_ctx_synchronize(gpu.ctx.handle))
            dt = (time.perf_counter() - t0)/n
            frames[tile] = gpu.read(dst, w, h, 3)
            line += f"  tile {tile}: {dt*1e3:8.3f} ms"
            N.check(gpu.lib.sfx_program_destroy(prog))
        if options.only:
            print(line, flush=True)
            continue
        assert frames["1"].mean() > 20 and frames["1"].std() > 0.2      # a picture (a wide blur of noise is nearly flat)
        print(line, " identical" if np.array_equal(frames["0"], frames["1"]) else " DIFFERENT", flush=True)
    gpu.close()
