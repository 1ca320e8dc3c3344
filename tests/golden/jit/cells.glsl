// Integer work: ivec2 coordinates, %, /, shifts and masks, unsigned hashing with wrap-around, switch, texelFetch and
// textureSize, ternaries, the prelude's palette and hsv2rgb.
uint hash(uint x) {
    x ^= x >> 16u;
    x *= 2246822519u;
    x ^= x >> 13u;
    x *= 3266489917u;
    x ^= x >> 16u;
    return x;
}

vec3 pick(int kind, float t) {
    vec3 colour;
    switch (kind) {
        case 0: colour = palette_magma(t); break;
        case 1: colour = hsv2rgb(vec3(t*TAU, 0.8, 0.9)); break;
        case 2: colour = vec3(t, 1.0 - t, 0.5); break;
        default: colour = vec3(t);
    }
    return colour;
}

void main() {
    ivec2 pixel = ivec2(fragCoord);
    ivec2 cell = pixel/8;
    ivec2 inside = pixel % 8;
    uint h = hash(uint(cell.x) + 977u*uint(cell.y) + uint(iFrame)*7919u);
    float t = float(h & 1023u)/1023.0;
    int kind = int((h >> 10u) % 4u);
    vec3 colour = pick(kind, t);
    ivec2 size = textureSize(background, 0);
    vec4 texel = texelFetch(background, ivec2(cell.x % size.x, cell.y % size.y), 0);
    colour = mix(colour, texel.rgb, 0.5);
    bool border = (inside.x == 0) || (inside.y == 0);
    colour *= border ? 0.5 : 1.0;
    if (((cell.x ^ cell.y) & 1) == 0) colour = colour.gbr;
    fragColor = vec4(colour, 1.0);
}
