"""
glsl2hip — fragments that are not in the kernel registry: GLSL 3.30 → HIP C++ → gfx950 code object.

The reference gives the assembled GLSL to the OpenGL driver (shader.py:190-239 builds the text, :313-349 compiles it
with `opengl.program`, falling back to `missing.glsl` on errors). Without a driver the same job is done in two steps:

1. `translate()` rewrites the fragment token by token into the body of a C++ struct that derives from
   `sf::rt::FragmentBase` (csrc/jit_runtime.hpp): globals and uniforms become members, functions become member
   functions, `main` becomes `main_`. GLSL and C++ share almost all of their expression and statement syntax; the
   header supplies the vector types (with swizzles), the built-in functions on the deterministic binary32 routines of
   csrc/sfmath.hpp and the reference's prelude, so what is rewritten is small:
     * floating literals get an `f` suffix (C++ would compute in double),
     * `in/out/inout` parameter qualifiers become values and references, precision/layout/interpolation qualifiers go,
     * `uniform`/`in`/`out` declarations go (the members exist already), prototypes go,
     * `int(x)`/`uint(x)` become `to_int(x)`/`to_uint(x)` (defined results for NaN and overflow),
     * array constructors `T[n](…)` become `{…}`, `T[n] name` becomes `T name[n]`,
     * `const` scalars with literal initialisers become `static constexpr` (array bounds),
     * `discard` sets a flag and returns, identifiers that are C++ keywords get a trailing underscore.
2. `compile()` runs hipcc (`--genco`, the flags of csrc/Makefile) and caches the code object by content hash;
   libshaderflow_hip loads it with `sfx_program_load`.

What is not supported raises `TranslationError` (the caller logs it and binds the `missing` kernel, like the reference
does for a GLSL compile error): geometry beyond one fullscreen quad, arrays as function
return values, uniform arrays, more than 16 samplers or 64 uniform floats, non-square matrices.
"""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
from dataclasses import dataclass, field
from pathlib import Path
from typing import Iterable, Optional

CSRC = Path(__file__).resolve().parent/"csrc"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
         "-fno-slp-vectorize", "-fno-gpu-flush-denormals-to-zero", "-fwrapv",      # GLSL integers wrap (§4.1.3); C++ leaves signed overflow undefined
         "-Wno-unused-value", "-Wno-deprecated-copy", "-Wno-parentheses"]

USER_SLOTS = 64          # csrc/glsl.hpp USER_SLOTS
TEX_SLOTS = 16           # csrc/glsl.hpp TEX_SLOTS
FIXED_SAMPLER_SLOTS = {"iSpectrogram": 1, "iWaveform": 2}      # the slots the tape patches per frame (render_kernels.hpp frame_view)


class TranslationError(Exception):
    pass


class CompileError(Exception):
    pass


@dataclass
class Binding:
    """One name the host sets on the program: a uniform (`slot` = first float of csrc Uniforms.user) or a sampler (`slot` = texture slot)"""
    name: str
    type: str
    slot: int
    count: int = 1
    integer: bool = False
    default: Optional[tuple] = None      # the initialiser of `uniform T name = …;` (GLSL 3.30 §4.3.5): what the program reads until the host sets the uniform
    array: Optional[str] = None          # element `name[i]` of the uniform array `array` (GL exposes array elements under these names too)

    @property
    def sampler(self) -> bool:
        return self.type == "sampler2D"


@dataclass
class Translation:
    cpp: str
    bindings: list[Binding] = field(default_factory=list)
    tiled_sampler: Optional[str] = None            # the sampler the kernels serve from an LDS tile (jit_runtime.hpp TileView), if any

    @property
    def key(self) -> str:
        return hashlib.sha256((self.cpp + runtime_fingerprint()).encode()).hexdigest()[:24]


# ---- tokens -----------------------------------------------------------------------------------------------------------

_TOKEN = re.compile(r"""
    (?P<comment>//[^\n]*|/\*.*?\*/)
  | (?P<pp>\#(?:[^\n\\]|\\.|\\\n)*)
  | (?P<number>0[xX][0-9a-fA-F]+[uU]?|(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?(?:lf|LF|[fFuU])?)
  | (?P<ident>[A-Za-z_]\w*)
  | (?P<op><<=|>>=|\+\+|--|<<|>>|<=|>=|==|!=|&&|\|\||\^\^|[-+*/%&|^]=|[-+*/%<>=!&|^~?:;,.(){}\[\]])
  | (?P<ws>\s+)
""", re.X | re.S)

_CPP_ONLY_KEYWORDS = {
    "new", "delete", "operator", "private", "protected", "friend", "virtual", "explicit", "mutable", "try", "catch", "throw",
    "typename", "auto", "register", "signed", "char", "wchar_t", "char8_t", "char16_t", "char32_t", "and", "or", "not", "xor", "bitand",
    "bitor", "compl", "and_eq", "or_eq", "xor_eq", "not_eq", "nullptr", "constexpr", "consteval", "constinit", "decltype", "noexcept",
    "static_assert", "thread_local", "alignas", "alignof", "concept", "requires", "co_await", "co_return", "co_yield", "export",
    "import", "module", "final", "override", "asm", "typeid", "dynamic_cast", "static_cast", "reinterpret_cast", "const_cast",
}
_DROPPED_QUALIFIERS = {"highp", "mediump", "lowp", "flat", "smooth", "noperspective", "centroid", "invariant", "precise"}
_TYPES = {"void", "float", "int", "uint", "bool", "vec2", "vec3", "vec4", "ivec2", "ivec3", "ivec4", "uvec2", "uvec3", "uvec4",
          "bvec2", "bvec3", "bvec4", "mat2", "mat3", "mat4", "mat2x2", "mat3x3", "mat4x4", "sampler2D"}
_UNIFORM_COUNTS = {"float": (1, False), "int": (1, True), "bool": (1, True), "uint": (1, True), "vec2": (2, False), "vec3": (3, False),
                   "vec4": (4, False), "ivec2": (2, True), "ivec3": (3, True), "ivec4": (4, True),
                   "uvec2": (2, True), "uvec3": (3, True), "uvec4": (4, True),
                   "mat2": (4, False), "mat3": (9, False), "mat4": (16, False)}
# members of sf::rt::FragmentBase (jit_runtime.hpp): uniforms and varyings that exist for every fragment
BUILTIN_MEMBERS = {
    "fragCoord", "stxy", "glxy", "stuv", "astuv", "gluv", "agluv", "gl_FragCoord", "fragColor", "instance",
    "iTime", "iTau", "iDuration", "iFrametime", "iDeltatime", "iCycle", "iAspectRatio", "iWidth", "iHeight", "iResolution", "iMouse",
    "iWantAspect", "iQuality", "iSSAA", "iFramerate", "iFrame", "iLayer", "iSubsample", "iRealtime", "iRendering", "iMouseInside",
    "iMouse1", "iMouse2", "iCameraMode", "iCameraProjection", "iCameraRight", "iCameraUpward", "iCameraForward", "iCameraPosition",
    "iCameraZenith", "iCameraSeparation", "iCameraZoom", "iCameraIsometric", "iCameraFocalLength", "iCameraOrbital", "iCameraDolly",
    "iAudioVolume", "iAudioVolumeIntegral", "iAudioSTD", "iSpectrogramLength", "iSpectrogramBins", "iSpectrogramScroll",
    "iWaveformLength", "iSpectrogramSmooth", "iSpectrogramOffset", "iSpectrogramMin", "iSpectrogramMax",
}


@dataclass
class Tok:
    kind: str
    text: str


def tokenize(source: str) -> list[Tok]:
    tokens, position = [], 0
    while position < len(source):
        match = _TOKEN.match(source, position)
        if match is None or (match.lastgroup == "pp" and source[source.rfind("\n", 0, position) + 1:position].strip()):
            raise TranslationError(f"unexpected character {source[position]!r} at offset {position}")
        tokens.append(Tok(match.lastgroup, match.group()))
        position = match.end()
    return tokens


def _float_literal(text: str) -> str:
    if text[:2] in ("0x", "0X") or text[-1] in "uU":
        return text
    if text.endswith(("lf", "LF")):
        text = text[:-2]
    if text[-1] in "fF":
        return text
    if ("." in text) or ("e" in text) or ("E" in text):
        return text + ("f" if "." in text or "e" in text.lower() else "")
    return text


def _matching(tokens: list[Tok], start: int, open_: str, close: str) -> int:
    """index of the token closing the bracket opened at `start`"""
    depth = 0
    for k in range(start, len(tokens)):
        if tokens[k].kind == "op":
            if tokens[k].text == open_:
                depth += 1
            elif tokens[k].text == close:
                depth -= 1
                if depth == 0:
                    return k
    raise TranslationError(f"unbalanced {open_!r}")


def _significant(tokens: list[Tok], start: int, step: int = 1) -> int:
    """next index from `start` (inclusive) in direction `step` that is not whitespace or a comment; len(tokens) / -1 if none"""
    k = start
    while 0 <= k < len(tokens) and tokens[k].kind in ("ws", "comment"):
        k += step
    return k


# ---- token-level rewrites that apply everywhere --------------------------------------------------------------------------

def _is_length_call(tokens: list[Tok], dot: int) -> bool:
    name = _significant(tokens, dot + 1)
    if name >= len(tokens) or tokens[name].text != "length":
        return False
    open_ = _significant(tokens, name + 1)
    close = _significant(tokens, open_ + 1) if open_ < len(tokens) else len(tokens)
    return open_ < len(tokens) and tokens[open_].text == "(" and close < len(tokens) and tokens[close].text == ")"


def _rewrite_tokens(tokens: list[Tok], structs: set[str]) -> list[Tok]:
    out: list[Tok] = []
    k = 0
    types = _TYPES | structs
    while k < len(tokens):
        t = tokens[k]
        if t.kind == "comment":
            out.append(Tok("ws", "\n"*t.text.count("\n") or " "))
        elif t.kind == "number":
            out.append(Tok("number", _float_literal(t.text)))
        elif t.kind == "op" and t.text == "^^":                       # logical exclusive or (§5.9): on bools that is `!=`
            # `!=` binds tighter than `&&`, `^^` looser: refuse the one spelling where that would change the meaning silently
            for step in (-1, 1):
                depth, j = 0, k + step
                while 0 <= j < len(tokens):
                    text = tokens[j].text if tokens[j].kind == "op" else ""
                    if text in ("(", "[") if step == 1 else text in (")", "]"):
                        depth += 1
                    elif text in (")", "]") if step == 1 else text in ("(", "["):
                        if depth == 0:
                            break
                        depth -= 1
                    elif depth == 0 and text in (";", ",", "?", ":", "=", "||", "{", "}"):
                        break
                    elif depth == 0 and text == "&&":
                        raise TranslationError("`&&` next to `^^` without parentheses: write (a && b) ^^ c")
                    j += step
            out.append(Tok("op", "!="))
        elif t.kind == "op" and t.text == "." and _is_length_call(tokens, k) and out and out[-1].kind == "ident":
            # `name.length()` of an array or vector (§4.1.9, §5.5) → length_of(name) (jit_runtime.hpp)
            name = out.pop()
            out.append(Tok("ident", f"length_of({name.text})"))
            k = _matching(tokens, _significant(tokens, _significant(tokens, k + 1) + 1), "(", ")")
        elif t.kind == "ident":
            nxt = _significant(tokens, k + 1)
            following = tokens[nxt].text if nxt < len(tokens) else ""
            if t.text in _DROPPED_QUALIFIERS:
                pass
            elif t.text == "layout" and following == "(":
                k = _matching(tokens, nxt, "(", ")")
            elif t.text in _CPP_ONLY_KEYWORDS:
                out.append(Tok("ident", t.text + "_"))
            elif t.text == "main":
                out.append(Tok("ident", "main_"))
            elif t.text in ("int", "uint") and following == "(":
                out.append(Tok("ident", "to_" + t.text))
            elif t.text in types and following == "[":
                # `T[n](a, b)` array constructor → `{a, b}`;  `T[n] name` → `T name[n]`
                close = _matching(tokens, nxt, "[", "]")
                after = _significant(tokens, close + 1)
                if after < len(tokens) and tokens[after].text == "(":
                    end = _matching(tokens, after, "(", ")")
                    inner = _rewrite_tokens(tokens[after + 1:end], structs)
                    out.append(Tok("op", "{")); out.extend(inner); out.append(Tok("op", "}"))
                    k = end
                elif after < len(tokens) and tokens[after].kind == "ident":
                    out.append(t); out.append(Tok("ws", " ")); out.append(Tok("ident", tokens[after].text))
                    out.extend(_rewrite_tokens(tokens[nxt:close + 1], structs))
                    k = after
                else:
                    out.append(t)
            else:
                out.append(t)
        else:
            out.append(t)
        k += 1
    return out


def _text(tokens: Iterable[Tok]) -> str:
    return "".join(t.text for t in tokens)


# ---- function heads ---------------------------------------------------------------------------------------------------

def _rewrite_parameters(tokens: list[Tok]) -> list[Tok]:
    """`in T a, out T b, inout T c, const in T d, T e[3], out T f[3]` → `T a, T& b, T& c, const T d, const T (&e)[3], T (&f)[3]`
    (an `in` array is a copy in GLSL; here it is a view that cannot be written, which a compiler error reports if a fragment does)"""
    out: list[Tok] = []
    parameter: list[Tok] = []

    def flush() -> None:
        words = [t for t in parameter if t.kind not in ("ws", "comment")]
        if not words:
            return
        reference = any(t.kind == "ident" and t.text in ("out", "inout") for t in words)
        constant = any(t.kind == "ident" and t.text == "const" for t in words)
        core = [t for t in words if not (t.kind == "ident" and t.text in ("in", "out", "inout", "const"))]
        bracket = next((i for i, t in enumerate(core) if t.kind == "op" and t.text == "["), None)
        if bracket is None:
            type_, rest = core[0], core[1:]
            text = ("const " if constant and not reference else "") + type_.text + ("&" if reference else "") + " " + _text(rest)
        else:
            if bracket < 2:
                raise TranslationError(f"unnamed array parameter: {_text(parameter).strip()!r}")
            type_, name, bounds = core[0], core[bracket - 1], core[bracket:]
            text = ("" if reference else "const ") + f"{type_.text} (&{name.text}){_text(bounds)}"
        if out:
            out.append(Tok("op", ", "))
        out.append(Tok("ident", text))

    depth = 0
    for t in tokens:
        if t.kind == "op" and t.text in "([":
            depth += 1
        elif t.kind == "op" and t.text in ")]":
            depth -= 1
        if t.kind == "op" and t.text == "," and depth == 0:
            flush(); parameter = []
        else:
            parameter.append(t)
    flush()
    return out


# ---- the global scope -------------------------------------------------------------------------------------------------

_CONSTANT_TOKENS = re.compile(r"^[-+*/%()\s\d.eEfFuUxXa-fA-F<>&|^~!?:]*$")


def _count_initialisers(tokens: list[Tok]) -> int:
    """number of top-level elements of the brace list starting at tokens[0] == '{'"""
    end = _matching(tokens, 0, "{", "}")
    depth, count, any_token = 0, 0, False
    for t in tokens[1:end]:
        if t.kind == "op" and t.text in "([{":
            depth += 1
        elif t.kind == "op" and t.text in ")]}":
            depth -= 1
        elif t.kind == "op" and t.text == "," and depth == 0:
            count += 1
        if t.kind not in ("ws",):
            any_token = True
    return count + 1 if any_token else 0


class _Translator:
    def __init__(self, source: str, uniforms: list[tuple[str, str]]):
        self.source = source
        self.pipeline = list(uniforms)
        self.body: list[str] = []
        self.structs: set[str] = set()
        self.constants: set[str] = set()
        self.constant_values: dict[str, int] = {}                    # `const int N = 3;` → array sizes of uniform declarations
        self.declared_uniforms: list[tuple[str, str, str]] = []     # (type, name, default text)
        self.macros: list[str] = []
        self.identifiers: set[str] = set()

    # a statement of the global scope that ends in ';'
    def declaration(self, tokens: list[Tok]) -> str:
        first = _significant(tokens, 0)
        if first >= len(tokens):
            return _text(tokens)
        words = [t.text for t in tokens if t.kind == "ident"]
        head = tokens[first].text
        if head == "precision":
            return ""
        qualifiers = set()
        k = first
        while k < len(tokens) and (tokens[k].kind in ("ws",) or (tokens[k].kind == "ident" and tokens[k].text in ("uniform", "in", "out", "varying", "attribute", "const"))):
            if tokens[k].kind == "ident":
                qualifiers.add(tokens[k].text)
            k += 1
        rest = tokens[k:]
        if "uniform" in qualifiers:
            type_index = _significant(rest, 0)
            type_ = rest[type_index].text
            names = _text(rest[type_index + 1:-1])
            for part in self._split_commas(names):
                name, _, default = part.partition("=")
                self.declared_uniforms.append((type_, name.strip(), default.strip()))
            return ""
        if qualifiers & {"in", "out", "varying", "attribute"}:
            type_index = _significant(rest, 0)
            names = [n.strip() for n in self._split_commas(_text(rest[type_index + 1:-1]))]
            unknown = [n for n in names if n.split("[")[0].strip() not in BUILTIN_MEMBERS]
            return (f"\n{rest[type_index].text} {', '.join(unknown)};" if unknown else "")
        # prototype: `T name(params);` without an initialiser
        texts = [t.text for t in tokens if t.kind not in ("ws",)]
        if "=" not in texts and "(" in texts and texts[-2] == ")" and len(words) >= 2:
            return ""
        if head == "struct":
            return self.struct(tokens)
        text = _text(tokens)
        if "const" in qualifiers:
            type_index = _significant(rest, 0)
            type_ = rest[type_index].text
            if type_ in ("int", "float", "bool", "uint") and "[" not in texts:
                parts = self._split_commas(_text(rest[type_index + 1:-1]))
                constant = True
                for part in parts:
                    name, _, init = part.partition("=")
                    probe = re.sub(r"\b(?:%s|true|false)\b" % "|".join(sorted(self.constants) or ["__none__"]), "1", init)
                    probe = re.sub(r"\bto_u?int\b|\bfloat\b|\bbool\b", "", probe)
                    constant = constant and bool(init.strip()) and bool(_CONSTANT_TOKENS.match(probe))
                if constant:
                    for part in parts:
                        self.constants.add(part.partition("=")[0].strip())
                        if type_ == "int" and part.partition("=")[2].strip().isdigit():
                            self.constant_values[part.partition("=")[0].strip()] = int(part.partition("=")[2])
                    return f"\nstatic constexpr {type_} {', '.join(p.strip() for p in parts)};"
        # members cannot deduce an array bound from their initialiser: write it out
        match = re.search(r"\[\s*\]\s*=\s*\{", text)
        if match:
            brace = next(i for i, t in enumerate(tokens) if t.kind == "op" and t.text == "{")
            text = text[:match.start()] + f"[{_count_initialisers(tokens[brace:])}] = {{" + text[match.end():]
        return text

    @staticmethod
    def _split_commas(text: str) -> list[str]:
        parts, depth, current = [], 0, ""
        for ch in text:
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            if ch == "," and depth == 0:
                parts.append(current); current = ""
            else:
                current += ch
        if current.strip():
            parts.append(current)
        return parts

    def struct(self, tokens: list[Tok]) -> str:
        return _text(tokens)

    def function(self, head: list[Tok], body: list[Tok]) -> str:
        open_ = next(i for i, t in enumerate(head) if t.kind == "op" and t.text == "(")
        close = _matching(head, open_, "(", ")")
        before = head[:open_]
        return_type = next((t.text for t in before if t.kind == "ident"), "void")
        parameters = _rewrite_parameters(head[open_ + 1:close])
        if _text(parameters).strip() == "void":
            parameters = []
        body_text = self.function_body(body, return_type)
        return "\nSF_HD " + _text(before).lstrip() + "(" + _text(parameters) + ")" + _text(head[close + 1:]) + body_text

    def function_body(self, body: list[Tok], return_type: str) -> str:
        out = []
        for t in body:
            if t.kind == "ident" and t.text == "discard":
                # the invocation ends in GLSL; here the flag makes the result transparent whatever the callers still compute
                out.append("{ discarded_ = true; return; }" if return_type == "void" else "{ discarded_ = true; return {}; }")
            elif t.kind == "pp":
                out.append("\n" + self.preprocessor(t.text).strip("\n") + "\n")
            else:
                out.append(t.text)
        return "".join(out)

    def preprocessor(self, text: str) -> str:
        stripped = text.strip()
        directive = re.match(r"#\s*(\w+)", stripped)
        name = directive.group(1) if directive else ""
        if name in ("version", "extension", "pragma", "line", "include"):
            return ""
        if name == "define":
            match = re.match(r"(\s*#\s*define\s+)(\w+)(.*)$", text, re.S)
            if match:
                self.macros.append(match.group(2))
                rewritten = _text(_rewrite_tokens(tokenize(match.group(3)), self.structs))
                return match.group(1) + (match.group(2) + "_" if match.group(2) in _CPP_ONLY_KEYWORDS else match.group(2)) + rewritten
        return text

    def run(self) -> Translation:
        raw = tokenize(self.source)
        # names the fragment can reach: those of its code, and those of the macros it reaches (a texture's `#define name
        # name0x0` must not make the sampler active unless `name` is used)
        self.identifiers = {t.text for t in raw if t.kind == "ident"}
        macros: dict[str, set[str]] = {}
        for t in raw:
            if t.kind == "pp":
                define = re.match(r"\s*#\s*define\s+(\w+)(?:\([^)]*\))?(.*)$", t.text, re.S)
                if define:
                    macros.setdefault(define.group(1), set()).update(re.findall(r"[A-Za-z_]\w*", define.group(2)))
                else:
                    self.identifiers |= set(re.findall(r"[A-Za-z_]\w*", t.text))
        grown = True
        while grown:
            grown = False
            for name, names in macros.items():
                if name in self.identifiers and not names <= self.identifiers:
                    self.identifiers |= names
                    grown = True
        # struct names first: they are types for the array rewrites
        for k, t in enumerate(raw):
            if t.kind == "ident" and t.text == "struct":
                nxt = _significant(raw, k + 1)
                if nxt < len(raw) and raw[nxt].kind == "ident":
                    self.structs.add(raw[nxt].text)
        tokens = _rewrite_tokens(raw, self.structs)
        k, statement = 0, []
        while k < len(tokens):
            t = tokens[k]
            if t.kind == "pp":
                self.body.append(_text(statement)); statement = []
                self.body.append("\n" + self.preprocessor(t.text).strip("\n") + "\n")
            elif t.kind == "op" and t.text == ";":
                statement.append(t)
                self.body.append(self.declaration(statement)); statement = []
            elif t.kind == "op" and t.text == "{":
                end = _matching(tokens, k, "{", "}")
                texts = [s.text for s in statement if s.kind != "ws"]
                if "=" in texts:                                   # brace initialiser of a global: part of the statement
                    statement.extend(tokens[k:end + 1])
                elif texts and texts[0] == "struct":               # struct definition (up to its ';')
                    statement.extend(tokens[k:end + 1])
                elif "(" in texts:                                 # function definition
                    self.body.append(self.function(statement, tokens[k:end + 1])); statement = []
                else:
                    raise TranslationError(f"unexpected block after {_text(statement).strip()!r}")
                k = end
            else:
                statement.append(t)
            k += 1
        if _text(statement).strip():
            raise TranslationError(f"unterminated declaration: {_text(statement).strip()[:60]!r}")
        return self.assemble()

    def _array_length(self, text: str) -> int:
        if text.isdigit():
            return int(text)
        value = self.constant_values.get(text)
        if value is None:
            raise TranslationError(f"array size '{text}' is not an integer literal or a literal `const int`")
        return int(value)

    def assemble(self) -> Translation:
        bindings: list[Binding] = []
        members: list[str] = []
        loads: list[str] = []
        seen: set[str] = set()
        next_float = 0
        free_samplers = [s for s in range(TEX_SLOTS) if s not in FIXED_SAMPLER_SLOTS.values()]
        wanted = [(type_, name, "") for (type_, name) in self.pipeline] + self.declared_uniforms
        declared_names = {name for (_, name, _) in self.declared_uniforms}
        for (type_, name, default) in wanted:
            if name in seen or name in BUILTIN_MEMBERS:
                continue
            if name not in self.identifiers and name not in declared_names:
                continue                                           # inactive uniform: the driver would have removed it as well
            seen.add(name)
            if type_ == "sampler2D":
                # the engine names a texture's newest box `<name>0x0` (texture.py:346-347) and #defines the plain name to it
                plain = name[:-3] if name.endswith("0x0") else name
                if plain in FIXED_SAMPLER_SLOTS:
                    slot = FIXED_SAMPLER_SLOTS[plain]
                elif free_samplers:
                    slot = free_samplers.pop(0)
                else:
                    raise TranslationError(f"more than {TEX_SLOTS} samplers")
                bindings.append(Binding(name, type_, slot))
                members.append(f"    sampler2D {name};")
                loads.append(f"{name} = sampler_({slot});")
                continue
            length = None                                          # `uniform T name[N]`: N elements, bound as name[0] … name[N-1]
            array = re.fullmatch(r"(\w+)\s*\[\s*(\w+)\s*\]", name)
            if array:
                name = array.group(1)
                length = self._array_length(array.group(2))
            elif "[" in name:
                raise TranslationError(f"uniform {name}: only one-dimensional arrays with a literal or constant size are supported")
            if type_ not in _UNIFORM_COUNTS:
                raise TranslationError(f"uniform {name}: type {type_} is not supported")
            count, integer = _UNIFORM_COUNTS[type_]
            if next_float + count*(length or 1) > USER_SLOTS:
                raise TranslationError(f"more than {USER_SLOTS} floats of uniforms")
            defaults = _constant_initialiser(type_, count, length, default, name) if default else None

            def load(target: str, first: int) -> str:
                getter = "user_int_" if integer else "user_"
                values = ", ".join(f"{getter}({first + i})" for i in range(count))
                if type_ == "bool":
                    return f"{target} = user_int_({first}) != 0;"
                if type_ == "uint":
                    return f"{target} = (uint)user_int_({first});"
                if count == 1:
                    return f"{target} = {values};"
                return f"{target} = {type_}({values});"

            if length is None:
                members.append(f"    {type_} {name};")
                loads.append(load(name, next_float))
                bindings.append(Binding(name, type_, next_float, count, integer, default=defaults[0] if defaults else None))
                next_float += count
            else:
                members.append(f"    {type_} {name}[{length}];")
                for index in range(length):
                    loads.append(load(f"{name}[{index}]", next_float))
                    bindings.append(Binding(f"{name}[{index}]", type_, next_float, count, integer,
                                            default=defaults[index] if defaults else None, array=name))
                    next_float += count
        code = "".join(self.body)
        undefs = "".join(f"#undef {m}\n" for m in dict.fromkeys(self.macros))
        derivatives = bool(self.identifiers & {'dFdx', 'dFdy', 'fwidth'})
        tiled = None if derivatives else _sampler_worth_a_tile(code, [b for b in bindings if b.type == "sampler2D"])
        cpp = ("// generated by shaderflow_amd/glsl2hip.py from a GLSL fragment\n"
               + (f"#define SF_JIT_TILE_SLOT {tiled.slot}      // {tiled.name}\n" if tiled else "") +
               "#include \"jit_runtime.hpp\"\n"
               "namespace sf { namespace rt {\n"
               "struct Fragment : FragmentBase {\n" + "\n".join(members) + "\n"
               "    SF_HD void load_user_() { " + " ".join(loads) + " }\n"
               "// ---- translated fragment ----\n" + code + "\n"
               "// ---- end of translated fragment ----\n"
               "};\n"
               "}}\n" + undefs +
               f"#define SF_JIT_DERIVATIVES {int(derivatives)}\n"
               "SF_JIT_ENTRY_POINTS(sf::rt::Fragment)\n")
        return Translation(cpp, bindings, tiled.name if tiled else None)


_TEXTURE_CALLS = r"(?:texture|textureLod|gtexture|gmtexture|stexture|astexture|agtexture|agmtexture)"


def _closing_in_text(code: str, at: int, opening: str, closing: str) -> int:
    """index just past the bracket that closes the one at `at`"""
    depth = 0
    for k in range(at, len(code)):
        if code[k] == opening:
            depth += 1
        elif code[k] == closing:
            depth -= 1
            if depth == 0:
                return k + 1
    return len(code)


def _loop_bodies(code: str) -> list[tuple[int, int]]:
    """(first, last) text ranges of the header and body of every for / while loop"""
    ranges = []
    for match in re.finditer(r"\b(?:for|while)\s*\(", code):
        header_end = _closing_in_text(code, match.end() - 1, "(", ")")
        rest = code[header_end:]
        body = header_end + len(rest) - len(rest.lstrip())
        if body < len(code) and code[body] == "{":
            end = _closing_in_text(code, body, "{", "}")
        else:
            semicolon = code.find(";", body)
            end = len(code) if semicolon < 0 else semicolon + 1
        ranges.append((match.start(), end))
    return ranges


def _sampler_worth_a_tile(code: str, samplers: list[Binding]) -> Optional[Binding]:
    """The sampler a tap-heavy fragment reads most (a blur or feedback kernel: taps inside the text of a loop, or eight and more
    written out), or None. SHADERFLOW_JIT_TILE=0 turns the tile off, =<sampler name> forces it for that sampler.

    The tile costs one extra evaluation of the fragment per block (jit_runtime.hpp JitShader::setup) and never changes a
    result, so the choice is about speed only: a single tap does not pay for the probe."""
    wish = os.environ.get("SHADERFLOW_JIT_TILE", "1")
    if wish.lower() in ("0", "off", "false", "no"):
        return None
    forced = [b for b in samplers if b.name == wish or (b.name.endswith("0x0") and b.name[:-3] == wish)]
    if forced:
        return forced[0]
    loops = _loop_bodies(code)
    in_loop = lambda at: any(first <= at < last for (first, last) in loops)
    weight = lambda at: 16 if in_loop(at) else 1
    tap_on = lambda names: re.compile(rf"\b{_TEXTURE_CALLS}\s*\(\s*(?:{'|'.join(map(re.escape, names))})\b")
    # helper functions that tap a sampler PARAMETER (`vec4 blur(sampler2D tex, vec2 uv) { for (…) … texture(tex, …) }`): the taps
    # count for whatever sampler a call passes in that position, once more heavily when the call itself sits in a loop
    helpers = []                                               # (name, position of the sampler parameter, weight of its taps)
    for definition in re.finditer(r"\bSF_HD\s+[\w:<>]+\s+(\w+)\s*\(([^()]*)\)\s*(?:const\s*)?\{", code):
        body_end = _closing_in_text(code, definition.end() - 1, "{", "}")
        for position, parameter in enumerate(_split_arguments(definition.group(2))):
            declared = re.fullmatch(r"\s*(?:const\s+)?sampler2D\s*&?\s*(\w+)\s*", parameter)
            if declared:
                taps = sum(weight(m.start()) for m in tap_on([declared.group(1)]).finditer(code, definition.end(), body_end))
                if taps:
                    helpers.append((definition.group(1), position, taps))
    best, best_score = None, 0
    for binding in samplers:
        names = {binding.name, binding.name[:-3] if binding.name.endswith("0x0") else binding.name}
        score = sum(weight(m.start()) for m in tap_on(names).finditer(code))
        for (helper, position, taps) in helpers:
            for call in re.finditer(rf"\b{re.escape(helper)}\s*\(", code):
                arguments = _split_arguments(code[call.end():_closing_in_text(code, call.end() - 1, "(", ")") - 1])
                if position < len(arguments) and arguments[position].strip() in names:
                    score += taps*weight(call.start())
        if score > best_score:
            best, best_score = binding, score
    return best if best_score >= 8 else None


_SCALAR_OF = {"float": float, "int": int, "uint": int, "bool": bool}


def _constant_scalar(text: str, name: str) -> float:
    """A literal arithmetic expression of a uniform initialiser → its value (`-0.5`, `2.0*3.0`, `1e-3`, `true`, `3u`)"""
    import ast
    cleaned = re.sub(r"(?<=[0-9.])(?:lf|LF|[fFuU])\b", "", text.strip())
    cleaned = re.sub(r"\btrue\b", "1", re.sub(r"\bfalse\b", "0", cleaned))
    cleaned = re.sub(r"\b(?:float|int|uint|bool)\s*\(", "(", cleaned)
    try:
        tree = ast.parse(cleaned, mode="eval")
    except SyntaxError:
        raise TranslationError(f"uniform {name}: initialiser '{text}' is not a literal constant expression") from None
    allowed = (ast.Expression, ast.BinOp, ast.UnaryOp, ast.Constant, ast.Add, ast.Sub, ast.Mult, ast.Div, ast.USub, ast.UAdd)
    if not all(isinstance(node, allowed) for node in ast.walk(tree)):
        raise TranslationError(f"uniform {name}: initialiser '{text}' is not a literal constant expression")
    import builtins                                            # this module defines its own compile()
    return float(eval(builtins.compile(tree, "<initialiser>", "eval"), {"__builtins__": {}}))


def _split_arguments(text: str) -> list[str]:
    parts, depth, current = [], 0, ""
    for ch in text:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(current)
            current = ""
        else:
            current += ch
    if current.strip():
        parts.append(current)
    return parts


def _constant_value(type_: str, count: int, text: str, name: str) -> tuple:
    """`T(…)` or a scalar expression → `count` numbers (vector constructors splat a single argument, matrices put it on the diagonal)"""
    text = text.strip()
    constructor = re.fullmatch(rf"{type_}\s*\((.*)\)", text, re.S)
    if count == 1:
        return (_constant_scalar(constructor.group(1) if constructor else text, name),)
    if not constructor:
        raise TranslationError(f"uniform {name}: initialiser '{text}' must be a {type_}(…) constructor of literals")
    arguments = [_constant_scalar(a, name) for a in _split_arguments(constructor.group(1))]
    if len(arguments) == 1:
        if type_.startswith("mat"):
            side = int(type_[3])
            return tuple(arguments[0] if (k % (side + 1) == 0) else 0.0 for k in range(count))
        return tuple(arguments*count)
    if len(arguments) != count:
        raise TranslationError(f"uniform {name}: {type_} initialiser with {len(arguments)} components")
    return tuple(arguments)


def _constant_initialiser(type_: str, count: int, length: Optional[int], text: str, name: str) -> list[tuple]:
    """The initialiser of a uniform declaration as one tuple per element (one element for a non-array)"""
    if length is None:
        return [_constant_value(type_, count, text, name)]
    # (array constructors reach this point already rewritten to C++ brace lists)
    array = re.fullmatch(rf"{type_}\s*\[\s*\w*\s*\]\s*\((.*)\)", text.strip(), re.S) or re.fullmatch(r"\{(.*)\}", text.strip(), re.S)
    if not array:
        raise TranslationError(f"uniform {name}: an array initialiser must be written {type_}[](…)")
    elements = _split_arguments(array.group(1))
    if len(elements) != length:
        raise TranslationError(f"uniform {name}: {len(elements)} initialisers for {length} elements")
    return [_constant_value(type_, count, element, name) for element in elements]


def translate(source: str, uniforms: Iterable[tuple[str, str]] = ()) -> Translation:
    """GLSL fragment text (defines + includes + content, without the `uniform` declarations the engine generates) and the
    pipeline's `(type, name)` pairs → C++ translation unit + the bindings of its uniforms and samplers"""
    return _Translator(source, list(uniforms)).run()


# ---- compile ----------------------------------------------------------------------------------------------------------

_fingerprint: Optional[str] = None


def runtime_fingerprint() -> str:
    """Hash of the headers a code object is compiled against (its RenderArgs layout must match the library's)"""
    global _fingerprint
    if _fingerprint is None:
        digest = hashlib.sha256()
        for name in ("sfmath.hpp", "glsl.hpp", "fragments.hpp", "render_kernels.hpp", "jit_runtime.hpp", "jit_swizzles.inc", "jit_intvec.inc",
                     "uniform_table.hpp"):
            digest.update((CSRC/name).read_bytes())
        digest.update(" ".join(FLAGS).encode())
        # …and the layout the LOADED library was built with (a variant library built with other switches, or headers edited since it
        # was built, must not pick up code objects cached for another layout; sfx_program_load checks the same value)
        try:
            from shaderflow_amd import _native
            digest.update(str(_native.lib().sfx_abi_layout()).encode())
        except Exception:                                      # no library yet (host-only translation tests): headers alone
            pass
        _fingerprint = digest.hexdigest()
    return _fingerprint


def cache_directory() -> Path:
    root = os.environ.get("SHADERFLOW_JIT_CACHE")
    if root:
        return Path(root)
    return Path(os.environ.get("XDG_CACHE_HOME", Path.home()/".cache"))/"shaderflow_amd"/"jit"


def compile(translation: Translation, *, cache: Optional[Path] = None, timeout: float = 300.0) -> bytes:
    """Translation → gfx950 code object (bytes), through the on-disk cache"""
    cache = Path(cache) if cache else cache_directory()
    cache.mkdir(parents=True, exist_ok=True)
    target = cache/f"{translation.key}.hsaco"
    if target.exists():
        return target.read_bytes()
    unit = cache/f"{translation.key}.hip"
    unit.write_text(translation.cpp)
    temporary = cache/f"{translation.key}.{os.getpid()}.tmp"
    command = [HIPCC, *FLAGS, f"-I{CSRC}", "--genco", str(unit), "-o", str(temporary)]
    try:
        done = subprocess.run(command, capture_output=True, text=True, timeout=timeout)
    except FileNotFoundError as error:
        raise CompileError(f"hipcc not found ({HIPCC}): fragments outside the registry need the ROCm compiler at run time") from error
    except subprocess.TimeoutExpired as error:
        raise CompileError(f"hipcc timed out after {timeout:.0f} s") from error
    if done.returncode != 0 or not temporary.exists():
        temporary.unlink(missing_ok=True)
        raise CompileError(f"hipcc failed ({unit}):\n{done.stderr[-4000:]}")
    os.replace(temporary, target)
    return target.read_bytes()
