"""
The native FLAC decoder (csrc/flac.inc) parses files a user hands to the scene: mutated streams must end in an error code, never in
an out-of-bounds access. tools/flac_fuzz.cpp compiles the decoder on its own with gcc under AddressSanitizer + UBSan and decodes
thousands of mutations (bit flips, splats, truncations, spliced runs, re-sealed header CRCs) of streams that exercise every subframe
kind; any sanitizer report aborts the harness. CPU only — the GPU pool has no sanitizer builds, and the decoder is host code anyway.
"""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from tests.flac_encoder import encode
from tests.test_host_flac import music

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    binary = tmp_path_factory.mktemp("flac_fuzz")/"flac_fuzz"
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            str(ROOT/"tools"/"flac_fuzz.cpp"), "-o", str(binary)], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert build.returncode == 0, build.stderr[-2000:]
    return binary


def seed_streams(directory: Path) -> list[Path]:
    """Three valid streams that between them use every subframe kind, Rice escapes, wasted bits, all channel assignments, 8/16/24
    bits and an unknown length in STREAMINFO"""
    lpc8 = ([1820, -1100, 420, -90, 30, -12, 5, -2], 12, 10)
    kinds = [dict(kind="lpc", order=8, lpc=lpc8, partition_order=3), dict(kind="fixed", order=2, wasted=2, partition_order=2, escape_partitions=(1,)),
             dict(kind="verbatim"), dict(kind="constant"), dict(kind="lpc", order=32, lpc=([3]*32, 5, 7), partition_order=1), dict(kind="fixed", order=4, wide=True)]
    paths = []
    for name, channels, bits, blocksize, known in (("stereo16", 2, 16, 576, True), ("mono24", 1, 24, 1152, False), ("stereo8", 2, 8, 256, True)):
        frames = 6
        pcm = (music(blocksize*frames, channels, bits, seed=bits + channels) >> 2) << 2            # two wasted bits everywhere
        for frame in range(frames):
            for c in range(channels):
                if kinds[(frame + c) % len(kinds)]["kind"] == "constant":
                    pcm[frame*blocksize:(frame + 1)*blocksize, c] = 4*(frame + 3)

        def plan(frame, count):
            # independent channels wherever a block was made constant (the side channel of a constant block is not)
            independent = any(kinds[(frame + c) % len(kinds)]["kind"] == "constant" for c in range(count))
            return dict(assignment=None if (count != 2 or independent) else (None, 8, 9, 10)[frame % 4],
                        subframes=[kinds[(frame + c) % len(kinds)] for c in range(count)])

        path = directory/f"{name}.flac"
        path.write_bytes(encode(pcm, 44100, bits, blocksize=blocksize, plan=plan, known_length=known))
        paths.append(path)
    return paths


def test_mutated_streams_never_trip_the_sanitizers(harness, tmp_path):
    streams = seed_streams(tmp_path)
    run = subprocess.run([str(harness), "30000", "20260214", *map(str, streams)], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "no sanitizer report" in run.stdout and "ERROR" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-3000:]
    decoded = int(run.stdout.split(" decoded")[0].split()[-1])
    assert decoded >= 1, run.stdout                                   # a few mutations still decode (padding, metadata): the harness reaches the decoder's end too
