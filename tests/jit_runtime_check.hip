// Host-side check of csrc/jit_runtime.hpp (compiled by tests/test_host_translate.py with hipcc --cuda-host-only): the GLSL
// semantics of the vector and matrix types that translated fragments rely on. Exits 0 when every identity holds.
#include "jit_runtime.hpp"

#include <cstdio>

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED line %d: %s\n", __LINE__, #cond); failures++; } } while (0)

// inside the namespace, like a translated fragment: the GLSL built-ins hide the C library's
namespace sf { namespace rt {
int run_checks() {
    // swizzles: reads, writes, compound assignment, aliasing
    vec4 v(1.0f, 2.0f, 3.0f, 4.0f);
    vec2 a = v.zx;
    CHECK(a.x == 3.0f && a.y == 1.0f);
    v.xy = v.yx;
    CHECK(v.x == 2.0f && v.y == 1.0f && v.z == 3.0f && v.w == 4.0f);
    v.wz = vec2(9.0f, 8.0f);
    CHECK(v.z == 8.0f && v.w == 9.0f);
    v.rgb *= 2.0f;
    CHECK(v.r == 4.0f && v.g == 2.0f && v.b == 16.0f && v.a == 9.0f);
    v.st += vec2(1.0f, 1.0f);
    CHECK(v.x == 5.0f && v.y == 3.0f);
    vec3 c = v.bgr;
    CHECK(c.x == 16.0f && c.y == 3.0f && c.z == 5.0f);
    vec4 w = vec4(c.zy, 1.0f, 2.0f);
    CHECK(w.x == 5.0f && w.y == 3.0f && w.z == 1.0f && w.w == 2.0f);
    CHECK(length(vec2(3.0f, 4.0f)) == 5.0f && dot(w.xy, w.xy) == 34.0f);
    CHECK((-w.xy).x == -5.0f && (2.0f*w.xy).y == 6.0f && (w.xy/w.yx).x == 5.0f/3.0f);
    CHECK(w.xyz == vec3(5.0f, 3.0f, 1.0f) && w.xy != w.yx);
    // constructors and int arguments (GLSL converts int to float implicitly)
    vec3 space = vec3(1, 11, 26)/255;
    CHECK(space.y == 11.0f/255.0f);
    CHECK(vec4(vec2(1.0f), vec2(2.0f)) == vec4(1.0f, 1.0f, 2.0f, 2.0f) && vec3(7.0f) == vec3(7.0f, 7.0f, 7.0f));
    CHECK(max(space.x, 0) == space.x && clamp(2.5f, 0, 1) == 1.0f && min(3, 4) == 3 && clamp(7, 0, 5) == 5 && abs(-3) == 3);
    CHECK(mix(vec3(0.0f), vec3(2.0f), 0.5f) == vec3(1.0f) && step(0.5f, vec2(0.25f, 0.75f)) == vec2(0.0f, 1.0f));
    CHECK(smoothstep(0, 1, 0.5f) == 0.5f && mod(vec2(5.0f, -1.0f), 2.0f) == vec2(1.0f, 1.0f));
    ivec2 p = ivec2(vec2(7.9f, -2.5f));
    CHECK(p.x == 7 && p.y == -2 && (p % 4).x == 3 && (p/2).x == 3);
    vec2 from_int = ivec2(3, 4);
    CHECK(from_int == vec2(3.0f, 4.0f));
    CHECK(to_int(0.0f/0.0f*0.0f + NAN) == 0 && to_int(1e20f) == 2147483647 && to_uint(-5.0f) == 0u && to_int(true) == 1);
    // matrices: column major constructors, both products, inverse
    mat2 m(1.0f, 2.0f, 3.0f, 4.0f);                                  // columns (1, 2) and (3, 4)
    CHECK(m[0] == vec2(1.0f, 2.0f) && m[1][0] == 3.0f);
    CHECK(m*vec2(1.0f, 1.0f) == vec2(4.0f, 6.0f) && vec2(1.0f, 1.0f)*m == vec2(3.0f, 7.0f));
    CHECK((m*m)[0] == vec2(7.0f, 10.0f) && transpose(m)[0] == vec2(1.0f, 3.0f) && determinant(m) == -2.0f);
    mat2 identity = m*inverse(m);
    CHECK(abs(identity[0].x - 1.0f) < 1e-6f && abs(identity[1].x) < 1e-6f);
    mat3 r3(vec3(0.0f, 1.0f, 0.0f), vec3(-1.0f, 0.0f, 0.0f), vec3(0.0f, 0.0f, 1.0f));
    CHECK(r3*vec3(1.0f, 0.0f, 0.0f) == vec3(0.0f, 1.0f, 0.0f) && determinant(r3) == 1.0f && inverse(r3)*vec3(0.0f, 1.0f, 0.0f) == vec3(1.0f, 0.0f, 0.0f));
    mat4 s4(2.0f);
    s4[3] = vec4(1.0f, 2.0f, 3.0f, 1.0f);
    CHECK(s4*vec4(1.0f, 1.0f, 1.0f, 1.0f) == vec4(3.0f, 4.0f, 5.0f, 1.0f) && determinant(s4) == 8.0f);
    CHECK(inverse(s4)*vec4(3.0f, 4.0f, 5.0f, 1.0f) == vec4(1.0f, 1.0f, 1.0f, 1.0f));
    vec2 q(1.0f, 0.0f);
    q *= rotate2d(0.0f);
    CHECK(q == vec2(1.0f, 0.0f));
    CHECK(mat3(s4)[2] == vec3(0.0f, 0.0f, 2.0f) && mat4(m)[1] == vec4(3.0f, 4.0f, 0.0f, 0.0f) && mat4(m)[3] == vec4(0.0f, 0.0f, 0.0f, 1.0f));
    vec3 turned(1.0f, 0.0f, 5.0f);
    turned.xy *= mat2(0.0f, 1.0f, -1.0f, 0.0f);                      // v*m: dot products with the columns
    CHECK(turned == vec3(0.0f, -1.0f, 5.0f));
    // integer vectors: unsigned wrap-around, shifts, swizzles, conversions
    uvec3 h = uvec3(1u, 2u, 3u)*1664525u + 1013904223u;
    h.x += h.y*h.z;
    h ^= h >> 16u;
    h = h.yzx + uvec3(4294967295u);                                  // wraps: minus one
    CHECK(h.x == ((2u*1664525u + 1013904223u) ^ ((2u*1664525u + 1013904223u) >> 16u)) - 1u);
    ivec3 n = ivec3(7, -3, 2) % 4;
    CHECK(n == ivec3(3, -3, 2) && (ivec3(8, 8, 8) >> 2) == ivec3(2) && (ivec2(5, 6) & 3) == ivec2(1, 2) && -ivec2(1, 2) == ivec2(-1, -2));
    ivec4 q4(ivec2(1, 2), ivec2(3, 4));
    q4.wx = q4.xw;
    CHECK(q4 == ivec4(4, 2, 3, 1) && q4.zy == ivec2(3, 2));
    CHECK(vec3(uvec3(1u, 2u, 3u)) == vec3(1.0f, 2.0f, 3.0f) && uvec2(vec2(3.9f, -1.0f)) == uvec2(3u, 0u) && ivec2(uvec2(7u, 8u)) == ivec2(7, 8));
    vec3 normalised = vec3(uvec3(4294967295u, 0u, 2147483648u))*(1.0f/4294967296.0f);
    CHECK(normalised.x == 1.0f && normalised.y == 0.0f && normalised.z == 0.5f);
    CHECK(floatBitsToUint(vec2(1.0f, -2.0f)) == uvec2(0x3f800000u, 0xc0000000u) && uintBitsToFloat(uvec2(0x3f800000u, 0x40000000u)) == vec2(1.0f, 2.0f));
    // relational
    CHECK(any(lessThan(vec3(1.0f, 2.0f, 3.0f), vec3(2.0f))) && !all(lessThan(vec3(1.0f, 2.0f, 3.0f), vec3(2.0f))) && all(not_(equal(vec2(1.0f), vec2(2.0f)))));
    vec2 whole;
    CHECK(modf(vec2(2.75f, -1.5f), whole) == vec2(0.75f, -0.5f) && whole == vec2(2.0f, -1.0f));
    CHECK(mix(vec3(1.0f), vec3(2.0f), lessThan(vec3(0.0f, 1.0f, 2.0f), vec3(1.0f))) == vec3(2.0f, 1.0f, 1.0f) && mix(1.0f, 2.0f, true) == 2.0f && mix(1.0f, 3.0f, 1) == 3.0f && mix(0, 1, 0.25f) == 0.25f);
    // prelude
    CHECK(stuv2gluv(vec2(0.5f)) == vec2(0.0f) && gluv2stuv(vec2(1.0f)) == vec2(1.0f) && zoom(vec2(1.0f), 2.0f) == vec2(4.0f));
    CHECK(sdSphere(vec3(0.0f), vec3(0.0f, 0.0f, 2.0f), 0.5f) == 1.5f && sdUnion(1.0f, 2.0f) == 1.0f && isBlackKey(1) && isWhiteKey(0.0f));
    CHECK(cmul(vec2(0.0f, 1.0f), vec2(0.0f, 1.0f)) == vec2(-1.0f, 0.0f) && cconj(vec2(1.0f, 2.0f)).y == -2.0f);
    CHECK(hsv2rgb(vec3(0.0f, 1.0f, 1.0f)) == vec3(1.0f, 0.0f, 0.0f) && rgb2hsv(vec3(1.0f, 0.0f, 0.0f)) == vec3(0.0f, 1.0f, 1.0f));
    CHECK(palette(0.0f, vec3(1.0f), vec3(2.0f), vec3(3.0f), vec3(4.0f)) == vec3(1.0f));
    if (failures == 0) std::printf("jit_runtime: all checks passed\n");
    return failures;
}
}}

int main() { return sf::rt::run_checks(); }
