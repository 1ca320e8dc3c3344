// Polar coordinates and the vector forms of the built-ins: atan2/atan1n, mod/fract/step/clamp/min/max on vectors,
// lessThan + any, distance/length, exp/log/sqrt/inversesqrt, sign/abs/floor/ceil, user uniforms of every type.
uniform float iSpin;
uniform float iScale;
uniform vec2 iCentre;
uniform int iRings;
uniform vec4 iTint;
uniform bool iInvert;

vec3 rings(vec2 p, float count) {
    float radius = length(p);
    float angle = atan2(p.y, p.x);
    float band = floor(radius*count);
    vec3 base = hsv2rgb(vec3(mod(angle + band*0.7 + iSpin, TAU), 0.7, 0.9));
    vec3 edge = smoothstep(vec3(0.0), vec3(0.08, 0.12, 0.16), vec3(fract(radius*count)));
    return base*edge;
}

void main() {
    vec2 p = gluv*iScale + iCentre;
    vec3 colour = rings(p, float(iRings));
    vec2 folded = abs(fract(p*2.0) - 0.5);
    vec2 stepped = step(vec2(0.25), folded);
    colour = mix(colour, colour.brg, 0.5*stepped.x*stepped.y);
    vec3 tone = clamp(min(colour*1.2, vec3(0.95)), vec3(0.05), max(colour, vec3(0.5)));
    tone += 0.1*vec3(exp(-length(p)), log(1.0 + distance(p, vec2(0.5))), sqrt(abs(p.x)))*inversesqrt(1.0 + dot(p, p));
    tone *= iTint.rgb*iTint.a;
    if (any(lessThan(tone, vec3(0.02)))) tone = vec3(0.02);
    if (iInvert) tone = 1.0 - tone;
    tone.r += 0.05*sign(p.x)*ceil(abs(atan1n(p))*4.0)/4.0;
    fragColor = vec4(tone, 1.0);
}
