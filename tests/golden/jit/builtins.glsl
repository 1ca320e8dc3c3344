// One built-in function per band of rows, evaluated over x in [-2, 2] (and y folded in where there are two arguments):
// pins the scalar and vector forms of GLSL 3.30 §8.1-8.5 that the other fragments do not reach.
float squash(float v) { return 0.5 + 0.25*clamp(v, -2.0, 2.0); }

void main() {
    float x = 4.0*astuv.x - 2.0;
    float unit = 2.0*astuv.x - 1.0;
    float positive = 0.05 + 2.0*astuv.x;
    int band = int(astuv.y*32.0);
    float t = fract(astuv.y*32.0);
    vec3 c = vec3(0.0);
    // two switches: the OpenGL ES implementation that renders the goldens drops the cases of one switch beyond the 24th
    if (band < 16) switch (band) {
        case 0: c = vec3(squash(tan(unit)), squash(asin(unit)), squash(acos(unit))); break;
        case 1: c = vec3(squash(sinh(x)), squash(cosh(x) - 2.0), squash(tanh(x))); break;
        case 2: c = vec3(squash(exp(x) - 2.0), squash(log(positive)), squash(exp2(x) - 2.0)); break;
        case 3: c = vec3(squash(log2(positive)), squash(sqrt(positive)), squash(inversesqrt(positive) - 1.0)); break;
        case 4: c = vec3(squash(floor(x)), squash(ceil(x)), squash(trunc(x))); break;
        case 5: c = vec3(squash(round(x + 0.25)), squash(roundEven(x + 0.25)), squash(sign(x)*abs(x))); break;
        case 6: c = vec3(squash(mod(x, 0.75)), squash(fract(x) - 0.5), squash(min(x, t) + max(x, -t))); break;
        case 7: c = vec3(squash(pow(positive, 1.5 + t) - 1.0), squash(atan(x)), squash(atan(x, t - 0.5))); break;
        case 8: c = vec3(squash(radians(90.0*x)), squash(degrees(x)/90.0), squash(mix(-1.0, 1.0, astuv.x))); break;
        case 9: c = vec3(step(0.3, x), smoothstep(-1.0, 1.5, x), squash(clamp(x, -0.5, 1.25))); break;
        case 10: c = vec3(squash(length(vec3(x, t, 0.5)) - 1.0), squash(distance(vec2(x, t), vec2(0.5, -0.5)) - 1.0), squash(dot(vec3(x, t, 1.0), vec3(0.5, -1.0, 0.25)))); break;
        case 11: c = 0.5 + 0.5*normalize(vec3(x, t - 0.5, 0.75)); break;
        case 12: c = 0.5 + 0.25*cross(vec3(x, t, 1.0), vec3(0.5, -0.25, t)); break;
        case 13: c = 0.5 + 0.25*reflect(vec3(x, -1.0, 0.5), normalize(vec3(0.2, 1.0, t))); break;
        case 14: c = 0.5 + 0.25*refract(normalize(vec3(unit, -1.0, 0.3)), vec3(0.0, 1.0, 0.0), 0.5 + t); break;
        case 15: c = 0.5 + 0.25*faceforward(vec3(0.3, 1.0, 0.2), vec3(unit, t - 0.5, 0.1), vec3(0.0, 1.0, 0.0)); break;
        default: break;
    } else switch (band) {
        case 16: c = 0.5 + 0.25*sin(vec3(x, 2.0*x, 3.0*x) + t); break;
        case 17: c = 0.5 + 0.25*cos(vec3(x, 2.0*x, 3.0*x) + t); break;
        case 18: c = exp2(-abs(vec3(x, x + t, x - t))); break;
        case 19: c = pow(vec3(positive, 0.5*positive, 0.25*positive), vec3(0.5, 1.5, 2.5))*0.3; break;
        case 20: c = mod(vec3(x, x + 0.5, x + 1.0), vec3(0.5, 0.75, 1.0)); break;
        case 21: c = mix(vec3(0.1, 0.8, 0.3), vec3(0.9, 0.2, 0.6), vec3(astuv.x, t, 0.5*(astuv.x + t))); break;
        case 22: c = clamp(vec3(x, -x, x*x), vec3(0.1), vec3(0.9, 0.8, 0.7)); break;
        case 23: c = smoothstep(vec3(-1.0, 0.0, 0.5), vec3(1.0, 1.5, 2.0), vec3(x)); break;
        case 24: c = vec3(lessThan(vec3(x), vec3(-1.0, 0.0, 1.0)))*0.5 + vec3(greaterThanEqual(vec3(t), vec3(0.25, 0.5, 0.75)))*0.25; break;
        case 25: c = vec3(any(greaterThan(vec2(x, t), vec2(1.0, 0.9))) ? 0.8 : 0.2, all(lessThanEqual(vec2(x, t), vec2(0.0, 0.5))) ? 0.7 : 0.3, any(not(equal(ivec2(int(x), 0), ivec2(0)))) ? 0.6 : 0.1); break;
        case 26: { mat3 m = mat3(1.0, 0.2, 0.0, -0.3, 1.0, 0.1, 0.5*x, t, 1.0); c = 0.5 + 0.2*(inverse(m)*vec3(1.0, 0.5, 0.25)) + 0.05*determinant(m); break; }
        case 27: { mat2 m = mat2(1.0 + t, x, -x, 1.0); vec2 r = transpose(m)*vec2(0.3, 0.6) + inverse(m)[1]; c = vec3(0.5 + 0.25*r, 0.5 + 0.1*determinant(m)); break; }
        case 28: { mat4 m = mat4(1.0); m[1] = vec4(x, 1.0, t, 0.0); m[3] = vec4(0.25, 0.5, 0.75, 1.0); vec4 r = inverse(m)*vec4(0.5, 0.25, 1.0, 1.0); c = 0.5 + 0.2*r.xyz*r.w + 0.02*determinant(m); break; }
        case 29: c = vec3(squash(float(abs(int(4.0*x)) % 3) - 1.0), squash(float(min(int(4.0*x), 2))), squash(float(clamp(int(4.0*x), -3, 1)))); break;
        case 30: c = vec3(isnan(sqrt(x)) ? 0.9 : 0.1, isinf(1.0/floor(abs(x))) ? 0.8 : 0.2, squash(x*t - 0.5)); break;
        default: c = vec3(squash(outerProduct(vec2(x, t), vec2(0.5, 0.25))[1].x), squash(matrixCompMult(mat2(x), mat2(t))[0].x), 0.5);
    }
    fragColor = vec4(c, 1.0);
}
