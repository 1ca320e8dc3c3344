// Screen-space derivatives: an anti-aliased ring whose edge width comes from fwidth, and dFdx/dFdy of smooth functions
// shown directly (scaled by the resolution so that they are of order one). The quadrant limits avoid pixel centres at odd sizes.
void main() {
    vec2 p = gluv*1.2;
    float ring = abs(length(p) - 0.6) - 0.15;
    float width = fwidth(ring);
    float coverage = 1.0 - smoothstep(-width, width, ring);
    float slope_x = dFdx(p.x*p.x)*iResolution.x*0.25;
    float slope_y = dFdy(sin(3.0*p.y))*iResolution.y*0.1;
    vec2 both = fwidth(vec2(p.x*p.y, p.x + p.y))*iResolution.y*0.2;
    vec3 colour = mix(vec3(0.1, 0.1, 0.2), vec3(0.9, 0.7, 0.3), coverage);
    colour = mix(colour, vec3(0.5 + 0.5*slope_x, 0.5 + 0.5*slope_y, both.x), step(0.0131, p.x)*step(0.0177, p.y));
    colour.b = mix(colour.b, both.y, step(p.x, -0.0131)*step(p.y, -0.0177));
    fragColor = vec4(clamp(colour, 0.0, 1.0), 1.0);
}
