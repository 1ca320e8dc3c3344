// A small sphere-traced scene through the scene camera: GetCamera, the prelude's distance functions, normals by
// central differences, reflect/normalize/cross, mat3 from columns, early return, break.
const int STEPS = 48;
const float FAR = 12.0;

float scene(vec3 p) {
    float ball = sdSphere(p, vec3(0.3, 0.1, 2.5), 0.6);
    float box = sdBox(p, vec3(-0.7, -0.2, 3.0), vec3(0.8));
    float ground = sdPlane(p, vec3(0.0, -0.8, 0.0), vec3(0.0, 1.0, 0.0));
    return sdUnion(sdSmoothUnion(ball, box, 0.3), ground);
}

vec3 normal_at(vec3 p) {
    vec2 e = vec2(0.002, 0.0);
    return normalize(vec3(scene(p + e.xyy) - scene(p - e.xyy),
                          scene(p + e.yxy) - scene(p - e.yxy),
                          scene(p + e.yyx) - scene(p - e.yyx)));
}

vec3 sky(vec3 direction) {
    return mix(vec3(0.6, 0.7, 0.9), vec3(0.1, 0.2, 0.5), clamp(direction.y*0.5 + 0.5, 0.0, 1.0));
}

void main() {
    GetCamera(iCamera);
    if (iCamera.out_of_bounds) {
        fragColor = vec4(0.0, 0.0, 0.0, 1.0);
        return;
    }
    vec3 origin = iCamera.origin;
    vec3 direction = normalize(iCamera.target - iCamera.origin);
    float travelled = 0.0;
    bool hit = false;
    for (int i = 0; i < STEPS; i++) {
        float d = scene(origin + direction*travelled);
        if (d < 0.003) { hit = true; break; }
        travelled += d;
        if (travelled > FAR) break;
    }
    if (!hit) {
        fragColor = vec4(sky(direction), 1.0);
        return;
    }
    vec3 p = origin + direction*travelled;
    vec3 n = normal_at(p);
    vec3 side = normalize(cross(n, vec3(0.0, 0.0, 1.0) + 0.001));
    mat3 basis = mat3(side, cross(n, side), n);
    vec3 light = normalize(basis*vec3(0.3, 0.4, 0.8) + vec3(0.4, 0.9, -0.5));
    float diffuse = max(dot(n, light), 0.0);
    float shine = pow(max(dot(reflect(direction, n), light), 0.0), 16.0);
    vec3 colour = vec3(0.9, 0.5, 0.3)*(0.15 + 0.85*diffuse) + vec3(shine);
    colour = mix(colour, sky(direction), smoothstep(0.0, FAR, travelled));
    fragColor = vec4(colour, 1.0);
}
