// Integer vectors: the pcg3d hash on uvec3 (wrap-around products, shifts, swizzled operands), ivec swizzle writes,
// uvec → vec conversion, bit casts, matrix-from-matrix constructors and a swizzle multiplied by a matrix in place.
uvec3 pcg3d(uvec3 v) {
    v = v*1664525u + 1013904223u;
    v.x += v.y*v.z;
    v.y += v.z*v.x;
    v.z += v.x*v.y;
    v ^= v >> 16u;
    v.x += v.y*v.z;
    v.y += v.z*v.x;
    v.z += v.x*v.y;
    return v;
}

vec3 random3(ivec3 cell) {
    uvec3 h = pcg3d(uvec3(cell));
    return vec3(h >> 8u)*(1.0/16777216.0);
}

void main() {
    ivec2 pixel = ivec2(fragCoord);
    ivec3 cell = ivec3(pixel/6, iFrame);
    cell.xy = cell.yx + ivec2(3, 5);
    vec3 colour = random3(cell);
    ivec3 turned = cell.zxy & 7;
    colour = mix(colour, vec3(turned)/7.0, 0.25);
    mat4 big = mat4(2.0);
    big[3] = vec4(0.25, 0.5, 0.75, 1.0);
    mat3 small = mat3(big);
    colour = 0.5*(small*colour)*0.5 + 0.25*colour;
    vec3 p = vec3(gluv, 0.0);
    p.xy *= mat2(0.0, 1.0, -1.0, 0.0);
    colour.b += 0.1*p.x;
    uint bits = floatBitsToUint(colour.r);
    colour.g = mix(colour.g, uintBitsToFloat((bits & 0xFFFF0000u)), 0.5);
    fragColor = vec4(clamp(colour, 0.0, 1.0), 1.0);
}
