#!/usr/bin/env python3
"""
Translate a GLSL fragment the way ShaderProgram.compile does for fragments outside the registry, without a GPU:
prints the bindings, writes the HIP C++ translation unit and (unless --no-compile) the gfx950 code object.

  python tools/translate.py shader.frag [-u float:iGain -u sampler2D:background …] [-o out_prefix] [--no-compile]

`-u type:name` declares a uniform the scene's pipeline would provide (built-in ones — iTime, iResolution, the camera, the audio
uniforms — need no declaration; `uniform` declarations inside the text are honoured as well).
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from shaderflow_amd import glsl2hip  # noqa: E402


def main() -> None:
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("fragment", type=Path)
    p.add_argument("-u", "--uniform", action="append", default=[], metavar="TYPE:NAME")
    p.add_argument("-o", "--output", type=Path, help="prefix of the files to write (default: next to the fragment)")
    p.add_argument("--no-compile", action="store_true")
    args = p.parse_args()
    uniforms = [tuple(item.split(":", 1)) for item in args.uniform]
    try:
        translation = glsl2hip.translate(args.fragment.read_text(), uniforms)
    except glsl2hip.TranslationError as error:
        raise SystemExit(f"translation error: {error}")
    prefix = args.output or args.fragment.with_suffix("")
    unit = Path(f"{prefix}.hip")
    unit.write_text(translation.cpp)
    print(f"{unit}: {len(translation.cpp.splitlines())} lines")
    for b in translation.bindings:
        print(f"  {'sampler' if b.sampler else 'uniform':8s} {b.type:10s} {b.name:24s} slot {b.slot}" + ("" if b.sampler else f" ({b.count} x {'int32' if b.integer else 'float'})"))
    if not args.no_compile:
        try:
            code = glsl2hip.compile(translation)
        except glsl2hip.CompileError as error:
            raise SystemExit(str(error))
        target = Path(f"{prefix}.hsaco")
        target.write_bytes(code)
        print(f"{target}: {len(code)} bytes (cache key {translation.key})")


if __name__ == "__main__":
    main()
