#!/usr/bin/env python3
"""
Where the oracle and the reference's llvmpipe frames of mesa_4k.npz (3840x2160, 2x SSAA, every 13th / 27th row) differ by more than
1 LSB, as coordinates → tests/golden/mesa_4k_outliers.npz ("store their coordinates and assert the set is
exactly those"). Needs only this repository (oracle + mesa_4k.npz), not /root/reference.

For both frames ("noise", "bench") and both filters of the oracle — "spec" (float weights, what the kernels compute) and "llvmpipe"
(its 8-bit fixed-point filter, sfo_set_llvmpipe_filter) — every (row, column, channel, |difference|) beyond 1 LSB, and per pixel
whether ONE of its four supersamples taking the colour of a supersample next to it (= sitting on the other side of an edge, a bar's
outline or the waveform strip, where the two implementations' last bits of atan/length decide) reproduces the reference's value.

Run:  python tests/golden/make_c3_outliers.py      (≈ 2 min on 8 cores)
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))

from oracle import binding as O  # noqa: E402
from tests.helpers import oracle_textures  # noqa: E402
from tests.test_oracle_mesa import c3_inputs, one_supersample_explains  # noqa: E402


def main() -> None:
    out = {}
    for name in ("noise", "bench"):
        K, u, arrays, params, w, h, ssaa = c3_inputs(name)
        textures = oracle_textures(arrays, params)
        for mode in ("spec", "llvmpipe"):
            found = []
            for n, r in enumerate(K[f"{name}.rows"]):
                r = int(r)
                lo = max(0, r*ssaa - 2)

                def go():
                    screen = O.render("visualizer", u, textures, w*ssaa, h*ssaa, rows=(lo, min(h*ssaa, (r + 1)*ssaa + 2)), threads=8)
                    return screen, O.resolve(screen, w, h, 2, rows=(r, r + 1))[r]
                if mode == "llvmpipe":
                    with O.llvmpipe_filter():
                        screen, want = go()
                else:
                    screen, want = go()
                got = K[f"{name}.final"][n]
                d = np.abs(got.astype(int) - want.astype(int))
                for x in np.unique(np.argwhere(d > 1)[:, 0]):
                    edge = one_supersample_explains(screen, r, int(x), got[x])
                    for c in np.flatnonzero(d[x] > 1):
                        found.append((r, int(x), int(c), int(d[x, c]), int(edge)))
            out[f"{name}.{mode}"] = np.array(found, np.int32).reshape(-1, 5)
            pixels = {(a, b) for a, b, *_ in found}
            print(f"{name:6s} {mode:9s}: {len(found)} values in {len(pixels)} pixels beyond 1 LSB (max {max(f[3] for f in found)}), "
                  f"{len({(a, b) for a, b, _, _, e in found if e})} pixels explained by one supersample on the other side of an edge")
    np.savez_compressed(HERE/"mesa_4k_outliers.npz", **out)


if __name__ == "__main__":
    main()
