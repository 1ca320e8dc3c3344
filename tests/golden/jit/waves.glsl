// Interfering plane waves over a sampled background: structs, array constructors, constants, macros, loops,
// in/out/inout parameters, matrix products on both sides, swizzle reads and writes, a mutable global.
#define WAVES 4
#define half_of(x) ((x)*0.5)
const float SPEED = 0.5;
const float PHASES[WAVES] = float[WAVES](0.0, 1.3, 2.1, 4.4);
const vec3 TINT = vec3(1.0, 0.8, 0.6);
float energy = 0.0;

struct Wave {
    vec2 direction;
    float frequency;
};

float height(Wave wave, vec2 point, float phase);

float height(Wave wave, vec2 point, float phase) {
    return sin(dot(wave.direction, point)*wave.frequency + iTime*SPEED + phase);
}

void accumulate(inout vec3 colour, in float amount, out float total) {
    colour.rg += amount*vec2(0.5, 0.25);
    colour.b -= half_of(amount);
    total = colour.r + colour.g;
    energy += total;
}

void main() {
    mat2 turn = rotate2d(0.3);
    vec2 uv = gluv*turn;
    uv = turn*uv + vec2(0.05, -0.02);
    float sum = 0.0;
    for (int i = 0; i < WAVES; i++) {
        float angle = float(i)*0.9;
        Wave wave = Wave(vec2(cos(angle), sin(angle)), 3.0 + float(i));
        sum += height(wave, uv, PHASES[i])/float(WAVES);
    }
    vec3 colour = 0.5 + 0.5*cos(TAU*(sum*0.5 + vec3(0.0, 0.33, 0.67)));
    float total;
    accumulate(colour, 0.2, total);
    accumulate(colour, 0.1, total);
    colour *= TINT*smoothstep(0.0, 2.0, total);
    colour = mix(colour, stexture(background, stuv + 0.02*vec2(sum, -sum)).rgb, 0.35);
    colour.xz = colour.zx;
    vec4 result = vec4(clamp(colour, 0.0, 1.0), 1.0);
    result.rgb *= (energy > 1.0) ? 1.0 : 0.8;
    fragColor = result;
}
