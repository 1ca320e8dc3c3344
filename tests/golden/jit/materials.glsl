// Control flow and aggregates: nested structs, arrays of structs, a global mutable array, while / do-while / continue,
// several returns, struct out-parameters, rotation matrices composed on both sides, value noise from integer lattices.
struct Material {
    vec3 albedo;
    float gloss;
};
struct Hit {
    float distance;
    int material;
    vec3 normal;
};
const int COUNT = 3;
const Material MATERIALS[COUNT] = Material[COUNT](
    Material(vec3(0.9, 0.3, 0.2), 8.0),
    Material(vec3(0.2, 0.7, 0.9), 32.0),
    Material(vec3(0.8, 0.8, 0.3), 2.0));
float visits[COUNT];

float lattice(ivec2 cell) {
    int n = cell.x*374761 + cell.y*668265;
    n = (n ^ (n >> 13))*1274126;
    return float((n ^ (n >> 16)) & 65535)/65535.0;
}

float value_noise(vec2 p) {
    ivec2 cell = ivec2(floor(p));
    vec2 f = fract(p);
    vec2 w = f*f*(3.0 - 2.0*f);
    float a = lattice(cell), b = lattice(cell + ivec2(1, 0)), c = lattice(cell + ivec2(0, 1)), d = lattice(cell + ivec2(1, 1));
    return mix(mix(a, b, w.x), mix(c, d, w.x), w.y);
}

float fbm(vec2 p) {
    float sum = 0.0, amplitude = 0.5;
    int octave = 0;
    mat2 turn = rotate2d(0.5);
    while (octave < 5) {
        octave++;
        if (octave == 3) { p = turn*p; continue; }
        sum += amplitude*value_noise(p);
        p = p*2.0*turn;
        amplitude *= 0.5;
    }
    return sum;
}

bool intersect(vec3 origin, vec3 direction, out Hit hit) {
    hit.distance = 1e9;
    hit.material = -1;
    for (int k = 0; k < COUNT; k++) {
        vec3 centre = vec3(float(k) - 1.0, 0.2*float(k), 3.0);
        vec3 oc = origin - centre;
        float b = dot(oc, direction);
        float c = dot(oc, oc) - 0.2;
        float disc = b*b - c;
        if (disc < 0.0) continue;
        float t = -b - sqrt(disc);
        if (t < 0.0 || t > hit.distance) continue;
        hit.distance = t;
        hit.material = k;
        hit.normal = normalize(origin + direction*t - centre);
        visits[k] += 1.0;
    }
    return hit.material >= 0;
}

vec3 shade(Hit hit, vec3 direction) {
    Material m = MATERIALS[hit.material];
    vec3 light = normalize(vec3(0.5, 0.8, -0.4));
    float diffuse = max(dot(hit.normal, light), 0.0);
    float shine = pow(max(dot(reflect(direction, hit.normal), light), 0.0), m.gloss);
    if (hit.material == 2) return m.albedo*(0.3 + 0.7*diffuse);
    return m.albedo*(0.2 + 0.8*diffuse) + 0.5*shine;
}

void main() {
    for (int k = 0; k < COUNT; k++) visits[k] = 0.0;
    vec3 direction = normalize(vec3(gluv, 1.5));
    Hit hit;
    vec3 colour;
    if (intersect(vec3(0.0, 0.1, 0.0), direction, hit)) {
        colour = shade(hit, direction);
    } else {
        float clouds = fbm(gluv*3.0 + 0.2*iTime);
        colour = mix(vec3(0.2, 0.3, 0.6), vec3(1.0), smoothstep(0.3, 0.8, clouds));
    }
    int steps = 0;
    float fade = 1.0;
    do {
        fade *= 0.97;
        steps++;
    } while (steps < 4 && visits[steps % COUNT] < 0.5);
    fragColor = vec4(colour*fade, 1.0);
}
