// fragments.hpp — the reference's fragment shaders restated as HIP device functions on glsl.hpp.
// There is no GLSL compiler at run time: `shader.fragment = path|str` resolves through a registry
// (capi: sfx_program_lookup) to one of these; anything unknown falls back to `missing`, which is what
// the reference does on a compile error (shader.py:323-340).
//
// `out vec4 fragColor` starts as vec4(0).
#pragma once

#include "glsl.hpp"

namespace sf {

enum : int {
    FRAG_DEFAULT = 0,       // resources/shaders/fragment/default.glsl:1-48
    FRAG_MISSING = 1,       // resources/shaders/fragment/missing.glsl:4-22
    FRAG_VISUALIZER = 2,    // examples/basic/shaders/visualizer.frag:6-74
    FRAG_BARS = 3,          // examples/basic/shaders/bars.frag:5-22
    FRAG_WAVEFORM = 4,      // examples/basic/shaders/waveform.frag:5-19
    FRAG_MULTI_CHILD = 5,   // examples/basic/demo.py:74-79
    FRAG_MULTI_MAIN = 6,    // examples/basic/demo.py:83-89
    FRAG_SHADERTOY = 7,     // examples/basic/shaders/shadertoy.frag:62-66
    FRAG_DYNAMICS = 8,      // examples/basic/demo.py:121-126
    FRAG_AUDIO = 9,         // examples/basic/demo.py:149-153
    FRAG_MULTIPASS = 10,    // examples/basic/shaders/multipass.frag:10-45   (layers = 2, demo.py:93-99)
    FRAG_MOTIONBLUR = 11,   // examples/basic/shaders/motionblur.frag:1-18   (temporal = 10, layers = 2, demo.py:103-110)
    FRAG_LIFE_SIMULATION = 12,  // examples/basic/shaders/life/simulation.glsl:7-54 (R32F, temporal = 10, demo.py:223-242)
    FRAG_LIFE_VISUALS = 13, // examples/basic/shaders/life/visuals.glsl:6-41
    FRAG_VIDEO = 14,        // examples/basic/shaders/video.frag:1-6
    FRAG_RAYMARCH = 15,     // examples/basic/shaders/raymarch.frag:5-59
    FRAG_MANDELBROT = 16,   // examples/fractals/shaders/mandelbrot.frag:1-30
    FRAG_TETRATION = 17,    // examples/fractals/shaders/tetration.frag:1-54
    FRAG_COUNT = 18,
};

// scene-defined uniforms of the fragments below, by user[] slot (capi: g_user_uniforms)
enum : int { USER_SCREEN_TEMPORAL = 0, USER_LIFE_SIZE = 0 /* vec2: 0, 1 */, USER_LIFE_PERIOD = 2 };

// ---- default.glsl ----------------------------------------------------------------------------------
SF_HD vec3 default_grid(vec2 uv, float grid) {
    if (sf::mod(::floorf(uv.x*grid/2.0f) + ::floorf(uv.y*grid/2.0f), 2.0f) > 0.5f) return {0.22f, 0.22f, 0.22f};
    return {0.20f, 0.20f, 0.20f};
}
SF_HD vec4 frag_default(const Frag& f) {
    const Uniforms& u = *f.u;
    Camera cam = get_camera(f);
    vec4 col = {0.0f, 0.0f, 0.0f, 0.0f};
    vec2 uv = cam.gluv;
    if (cam.out_of_bounds) return {0.15f, 0.15f, 0.15f, 1.0f};
    float angle = sf::atan2(uv.y, uv.x);
    vec3 color = hsv2rgb(angle + (2.0f*TAU*u.iTau) - (PI/4.0f), 1.0f, 1.0f) + 0.3f;
    float circle = (1.333f*length(uv) - 1.0f);
    float width = 2.0f*sf::abs(1.0f/(circle*circle))*1e-4f;
    if (circle < 0.0f) set_rgb(col, rgb(col) + 0.18f);
    else set_rgb(col, rgb(col) + default_grid(uv, 8.0f));
    set_rgb(col, rgb(col) + (width*color));
    col.w = 1.0f;
    vec2 away = f.astuv*vec2{1.0f - f.astuv.y, 1.0f - f.astuv.x};
    float linear = 50.0f*(away.x*away.y);
    set_rgb(col, rgb(col)*sf::clamp(sf::pow(linear, 0.1f), 0.0f, 1.0f));
    return col;
}

// ---- missing.glsl ----------------------------------------------------------------------------------
SF_HD vec4 frag_missing(const Frag& f) {
    vec4 col = {0.0f, 0.0f, 0.0f, 0.0f};
    vec2 uv = f.stuv + vec2{f.u->iTime, f.u->iTime}/64.0f;
    const float size = 8.0f;
    for (int x = -5; x < 5; x++) {
        for (int y = -5; y < 5; y++) {
            vec2 block = {::floorf(size*uv.x), ::floorf(size*uv.y)};
            if (sf::mod(block.x + block.y, 2.0f) == 0.0f) {
                col.x += 1.0f/25.0f; col.y += 0.0f/25.0f; col.z += 1.0f/25.0f;
            }
        }
    }
    col.w = 0.2f;
    return col;
}

// ---- visualizer.frag ---------------------------------------------------------------------------------
// Split in three so that the LDS-tiled kernel can substitute its own evaluation of the radial blur (:21-33):
//   visualizer_consts  everything that depends on uniforms only (evaluated once per frame on the fast path,
//                      per fragment on the generic path: same operations, same bits),
//   visualizer_pre     camera + background coordinate of the fragment (:7-18),
//   visualizer_post    everything after the blur (:36-73).
struct VisualizerConsts {
    float zoom2;            // z*z, z = 0.95 + 0.01*sin(iTime) - 0.02*iAudioVolume - 0.03      (:17, shaderflow.glsl:361-363)
    float off_x, off_y;     // 0.005*cos(iTime*3.25135), 0.005*sin(iTime*1.153469)             (:18)
    float intensity;        // 0.01*clamp(pow(iAudioVolume, 2.5), 0, 0.3)                      (:22)
    float flash;            // 5*iAudioSTD                                                     (:36)
    float rot_c, rot_s;     // cos(-PI/2), sin(-PI/2)                                          (:39)
    float shrink;           // 1 - 0.4*pow(abs(iAudioVolume), 0.5)                             (:40)
    float vig_exp;          // 0.1 + 0.15*iAudioVolume                                         (:66)
};

SF_HD VisualizerConsts visualizer_consts(float iTime, float iAudioVolume, float iAudioSTD) {
    VisualizerConsts c;
    const float z = 0.95f + 0.01f*sf::sin(iTime) - 0.02f*iAudioVolume - 0.03f;
    c.zoom2 = z*z;
    c.off_x = 0.005f*sf::cos(iTime*3.25135f);
    c.off_y = 0.005f*sf::sin(iTime*1.153469f);
    c.intensity = 0.01f*sf::clamp(sf::pow(iAudioVolume, 2.5f), 0.0f, 0.3f);
    c.flash = 5.0f*iAudioSTD;
    c.rot_c = sf::cos(-PI/2.0f);
    c.rot_s = sf::sin(-PI/2.0f);
    c.shrink = 1.0f - 0.4f*sf::pow(sf::abs(iAudioVolume), 0.5f);
    c.vig_exp = 0.1f + 0.15f*iAudioVolume;
    return c;
}

struct VisualizerPre {
    vec2 uv, bg;            // iCamera.gluv, background_uv
    bool out_of_bounds;
};

SF_HD VisualizerPre visualizer_pre(const Frag& f, const VisualizerConsts& c, bool identity_camera = false) {
    VisualizerPre p;
    if (identity_camera) {                                         // glsl.hpp camera_is_identity(): same bits, no ray maths
        p.uv = f.gluv;
        p.out_of_bounds = (sf::abs(f.gluv.x) > f.u->iWantAspect);  // t == 1 (camera.glsl:83)
    } else {
        Camera cam = get_camera(f);
        p.uv = cam.gluv;
        p.out_of_bounds = cam.out_of_bounds;
    }
    p.bg = (gluv2stuv(p.uv) - vec2{0.5f, 0.5f})*c.zoom2 + vec2{0.5f, 0.5f};                   // zoom(uv, z, vec2(0.5)) :17
    p.bg = p.bg + vec2{c.off_x, c.off_y};                                                      // :18
    return p;
}

// :19-33 as written: centre tap + 9 directions x 10 steps (the float loop counters are binary32)
SF_HD vec4 visualizer_blur_reference(const Frag& f, const VisualizerPre& p, const VisualizerConsts& c) {
    const Tex& background = f.tex[TEX_BACKGROUND];
    vec4 color = stexture(background, p.bg);
    const float quality = 10.0f, directions = 8.0f;
    for (float angle = 0.0f; angle < TAU; angle += TAU/directions) {
        for (float walk = 1.0f/quality; walk <= 1.001f; walk += 1.0f/quality) {
            vec2 displacement = vec2{sf::cos(angle), sf::sin(angle)}*walk*c.intensity;
            color = color + stexture(background, p.bg + displacement);
        }
    }
    return color/(quality*directions);
}

// COLOUR_ONLY_FAST: the LDS-tiled kernel evaluates the four terms that only scale colours (never decide a branch or
// a texel index) with the hardware's v_log/v_exp/v_sqrt/v_rcp (1 ulp) instead of the sfmath sequences: <= 1e-6
// relative on a colour, far inside the 1 LSB tolerance of that path. Everything that feeds a comparison or a
// nearest-texel lookup (atan, the bar height, the lengths compared with it) stays bit-identical to the generic chain.
template <bool COLOUR_ONLY_FAST>
struct ColourMath {
    SF_HD static float pow(float x, float y) { return sf::pow(x, y); }
    SF_HD static float pow6(float x) { return sf::pow(x, 6.0f); }
    SF_HD static float length(vec2 a) { return sf::length(a); }
    SF_HD static float over20(float x) { return x/20.0f; }
};
#if defined(__HIP_DEVICE_COMPILE__)
template <> struct ColourMath<true> {
    __device__ __forceinline__ static float pow(float x, float y) {
        if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : INFINITY);      // as sf::pow
        return __builtin_amdgcn_exp2f(y*__builtin_amdgcn_logf(x));                      // v_exp_f32(y*v_log_f32(x)); x < 0 gives NaN like sf::pow
    }
    __device__ __forceinline__ static float pow6(float x) { const float x2 = x*x; return x2*x2*x2; }
    __device__ __forceinline__ static float length(vec2 a) { return __builtin_amdgcn_sqrtf(a.x*a.x + a.y*a.y); }
    __device__ __forceinline__ static float over20(float x) { return x*0.05f; }
};
#endif

template <bool COLOUR_ONLY_FAST = false>
SF_HD vec4 visualizer_post(const Frag& f, const VisualizerPre& p, const VisualizerConsts& c, vec4 col, const float* bars = nullptr) {
    using CM = ColourMath<COLOUR_ONLY_FAST>;
    const vec3 space = vec3{1.0f, 11.0f, 26.0f}/255.0f;                                        // :9
    col = col*(1.0f + c.flash*CM::pow6(sf::clamp(CM::length(f.agluv) - 0.3f, 0.0f, 1.0f)));   // :36

    vec2 music_uv = {c.rot_c*p.uv.x + c.rot_s*p.uv.y, (-c.rot_s)*p.uv.x + c.rot_c*p.uv.y};     // rotate2d(-PI/2)*uv :39
    music_uv = music_uv*c.shrink;                                                              // :40
    const float radius = 0.17f;

    float circle = sf::abs(atan1n(music_uv));                                                  // :44
    vec2 freq;
    if (bars) {
        // the nearest texel texture() would pick in the one-column RG32F spectrogram (glsl.hpp texture_xy: u = 0, clamp in y),
        // read from the per-frame table of sqrt(texel/1000) (k_visualizer_bars): the same bits, evaluated once per texel
        const Tex& sp = f.tex[TEX_SPECTROGRAM];
        const int j = wrap_texel((int)::floorf(circle*(float)sp.height), sp.height, sp.repeat_y);
        freq = {bars[2*j], bars[2*j + 1]};
    } else {
        vec2 s = texture_xy(f.tex[TEX_SPECTROGRAM], vec2{0.0f, circle});
        freq = {sf::sqrt(s.x/1000.0f), sf::sqrt(s.y/1000.0f)};                                // :45
    }
    freq = freq*(0.05f + 3.0f*sf::smoothstep(0.0f, 2.0f, circle));                            // :46

    float len = length(music_uv);
    if (len < radius) {                                                                        // :49-50
        set_rgb(col, rgb(col)*0.5f);
    } else {
        float bar = (music_uv.y < 0.0f) ? freq.x : freq.y;                                     // :52
        float r = radius + 0.5f*bar;
        if (len < r) {
            set_rgb(col, mix(rgb(col), vec3{1.0f, 1.0f, 1.0f}, sf::smoothstep(0.0f, 1.0f, 0.5f + bar)));   // :56
        } else {
            set_rgb(col, rgb(col)*CM::pow((len - r)*0.5f, 0.05f));                             // :58
        }
    }
    set_rgb(col, mix(rgb(col), space, sf::smoothstep(0.0f, 1.0f, CM::over20(CM::length(p.uv)))));   // :62

    vec2 vig = f.astuv*vec2{1.0f - f.astuv.y, 1.0f - f.astuv.x};                               // :65
    set_rgb(col, rgb(col)*CM::pow(vig.x*vig.y*20.0f, c.vig_exp));                              // :66
    col.w = 1.0f;

    vec2 w = texture_xy(f.tex[TEX_WAVEFORM], vec2{f.astuv.x, 0.0f});                           // :71
    vec2 wave = {0.2f*w.x, 0.2f*w.y};
    if (1.0f - f.gluv.y < wave.x) col = col*0.8f;
    if (1.0f + f.gluv.y < wave.y) col = col*0.8f;
    return col;
}

SF_HD vec4 frag_visualizer(const Frag& f) {
    const VisualizerConsts c = visualizer_consts(f.u->iTime, f.u->iAudioVolume, f.u->iAudioSTD);
    VisualizerPre p = visualizer_pre(f, c);
    if (p.out_of_bounds) {                                                                     // :11-14
        const vec3 space = vec3{1.0f, 11.0f, 26.0f}/255.0f;
        return {space.x, space.y, space.z, 0.0f};
    }
    return visualizer_post<false>(f, p, c, visualizer_blur_reference(f, p, c));
}

// ---- bars.frag / waveform.frag ---------------------------------------------------------------------
SF_HD vec4 frag_bars(const Frag& f) {
    vec4 col = {0.0f, 0.0f, 0.0f, 0.0f};
    vec2 s = texture_xy(f.tex[TEX_SPECTROGRAM], vec2{f.astuv.y, f.astuv.x});
    vec2 intensity = {sf::sqrt(s.x)/120.0f, sf::sqrt(s.y)/120.0f};
    if (f.astuv.y < intensity.x) { col.x += 1.0f; col.y += 0.0f; col.z += 0.0f; }
    if (f.astuv.y < intensity.y) { col.x += 0.0f; col.y += 1.0f; col.z += 0.0f; }
    if (f.astuv.y < (intensity.y + intensity.x)/2.0f) { col.x += 0.0f; col.y += 0.0f; col.z += 1.0f; }
    col.z += 0.4f*(intensity.x + intensity.y)*(1.0f - f.astuv.y);
    col.w = 1.0f;
    return col;
}
SF_HD vec4 frag_waveform(const Frag& f) {
    vec2 w = texture_xy(f.tex[TEX_WAVEFORM], vec2{f.astuv.x, 0.0f});
    vec4 col = {0.2f, 0.2f, 0.2f, 1.0f};
    float ay = sf::abs(f.gluv.y);
    if (ay < w.x) col.x = 1.0f;
    if (ay < w.y) col.y = 1.0f;
    if (ay < (w.x + w.y)/2.0f) col.z = 1.0f;
    return col;
}

// ---- small inline scenes of examples/basic/demo.py ---------------------------------------------------
SF_HD vec4 frag_multi_child(const Frag& f) { return {0.0f, 1.0f - f.stuv.x, 0.0f, 1.0f}; }
SF_HD vec4 frag_multi_main(const Frag& f) {
    vec4 c = texture(f.tex[TEX_CHILD], f.astuv);
    return {f.stuv.x + c.x, 0.0f + c.y, 0.0f + c.z, 1.0f};
}
SF_HD vec4 frag_shadertoy(const Frag& f) {
    float t = f.u->iTime;
    return {0.5f + 0.5f*sf::cos(t + f.stuv.x + 0.0f), 0.5f + 0.5f*sf::cos(t + f.stuv.y + 2.0f),
            0.5f + 0.5f*sf::cos(t + f.stuv.x + 4.0f), 1.0f};
}
SF_HD vec4 frag_dynamics(const Frag& f) {
    return stexture(f.tex[TEX_BACKGROUND], zoom(f.stuv, 0.85f + 0.1f*f.u->user[0], vec2{0.5f, 0.5f}));
}
SF_HD vec4 frag_audio(const Frag& f) {
    float v = f.u->iAudioVolume;
    return {v, v, v, 1.0f};
}

// ---- multipass.frag ---------------------------------------------------------------------------------
// blur() :10-26 with its float loop counters; `radius` arrives as float 5, `directions` and `steps` as int 8
SF_HD vec4 multipass_blur(const Tex& image, vec2 stuv, float radius, int directions, int steps) {
    vec4 color = {0.0f, 0.0f, 0.0f, 0.0f};
    float weights = 0.0f;
    for (float direction = 0.0f; direction < TAU; direction += TAU/(float)directions) {
        for (float walk = 1.0f/(float)steps; walk < 1.0f; walk += 1.0f/(float)steps) {
            vec2 offset = vec2{sf::cos(direction), sf::sin(direction)}*radius*walk/2000.0f;
            vec4 sample = texture(image, stuv + offset);
            float weight = 1.0f - length(offset - vec2{0.0f, 0.0f})/radius;
            color = color + sample*weight;
            weights += weight;
        }
    }
    return color/weights;
}
SF_HD vec4 frag_multipass(const Frag& f) {
    vec4 col = {0.0f, 0.0f, 0.0f, 0.0f};
    if (f.u->iLayer == 0) {
        col = stexture(f.tex[TEX_BACKGROUND], f.stuv);                                         // :32
    } else if (f.u->iLayer == 1) {
        const Tex& first = f.history[0];                                             // iScreen0x0
        col = texture(first, f.astuv);                                                         // :35
        if (f.gluv.x < 0.0f) col.x = 1.0f - col.x;                                             // :38-39
        else col = multipass_blur(first, f.astuv, 5.0f, 8, 8);                                 // :41
    }
    col.w = 1.0f;
    return col;
}

// ---- motionblur.frag --------------------------------------------------------------------------------
SF_HD vec4 frag_motionblur(const Frag& f) {
    vec4 col = {0.0f, 0.0f, 0.0f, 0.0f};
    if (f.u->iLayer == 0) {
        Camera cam = get_camera(f);
        col = stexture(f.tex[TEX_BACKGROUND], cam.stuv);                                       // :5-6
    } else if (f.u->iLayer == 1) {
        const int temporal = (int)f.u->user[USER_SCREEN_TEMPORAL];                             // iScreenTemporal
        vec4 color = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = 0; i < temporal; i++) {
            float factor = sf::smoothstep(1.0f, 0.0f, (float)i/(float)temporal);               // :11
            color = color + texture(f.history[i], f.astuv)*factor;                             // iScreenTexture(i, 0, astuv) :12
        }
        col = (color*2.0f)/(float)temporal;                                                    // :14
    }
    col.w = 1.0f;
    return col;
}

// ---- life/simulation.glsl, life/visuals.glsl ---------------------------------------------------------------
SF_HD vec4 frag_life_simulation(const Frag& f) {
    const Tex& previous = f.history[1];                                              // iLife1x0
    const int period = (int)f.u->user[USER_LIFE_PERIOD];
    vec4 col = {0.0f, 0.0f, 0.0f, 0.0f};
    col.w = 1.0f;
    if ((f.u->iFrame % period) != 0) {                                                         // :26-30
        col.x = texture(previous, f.astuv).x;
        return col;
    }
    const int px = sf::to_int(f.astuv.x*f.u->user[USER_LIFE_SIZE]), py = sf::to_int(f.astuv.y*f.u->user[USER_LIFE_SIZE + 1]);   // :32
    int near = 0, current = 0;
    for (int x = -1; x <= 1; x++) {
        for (int y = -1; y <= 1; y++) {
            const int cell = (texel_fetch(previous, px + x, py + y).x > 0.5f) ? 1 : 0;        // :38
            if (x == 0 && y == 0) current = cell; else near += cell;
        }
    }
    // alive[] :8-12 survives with two or three neighbours, dead[] :15-19 is born with three
    const bool lives = (current == 1) ? (near == 2 || near == 3) : (near == 3);
    col.x = lives ? 1.0f : 0.0f;
    return col;
}
SF_HD vec4 frag_life_visuals(const Frag& f) {
    const vec3 c1 = {0.01060815f, 0.01808215f, 0.10018654f}, c2 = {0.38092887f, 0.12061482f, 0.32506528f};
    const vec3 c3 = {0.79650140f, 0.10506637f, 0.31063031f}, c4 = {0.95922872f, 0.53307513f, 0.37488950f};
    Camera cam = get_camera(f);
    if (cam.out_of_bounds) return {c1.x, c1.y, c1.z, 1.0f};                                    // :15-18
    const float exponent = 1.3f;
    const float area = 1.0f/(exponent + 1.0f);
    float life = 0.0f;
    life += stexture(f.history[0], cam.stuv).x;                                      // :27-31
    life += stexture(f.history[1], cam.stuv).x*sf::pow(0.8f, exponent);
    life += stexture(f.history[2], cam.stuv).x*sf::pow(0.6f, exponent);
    life += stexture(f.history[3], cam.stuv).x*sf::pow(0.4f, exponent);
    life += stexture(f.history[4], cam.stuv).x*sf::pow(0.2f, exponent);
    life /= (5.0f*area);
    vec3 c = palette(life, c1, c2, c3, c4);
    return {c.x, c.y, c.z, 1.0f};
}

// ---- video.frag -------------------------------------------------------------------------------------
SF_HD vec4 frag_video(const Frag& f) {
    Camera cam = get_camera(f);
    vec4 col = stexture(f.history[0], cam.stuv);                                     // iVideo = iVideo0x0
    col.w = 1.0f;
    return col;
}

// ---- raymarch.frag ----------------------------------------------------------------------------------
SF_HD float sd_box(vec3 origin, vec3 point, vec3 size) {                                        // shaderflow.glsl:290-293
    vec3 d = {sf::abs(origin.x - point.x) - size.x/2.0f, sf::abs(origin.y - point.y) - size.y/2.0f, sf::abs(origin.z - point.z) - size.z/2.0f};
    vec3 m = {sf::max(d.x, 0.0f), sf::max(d.y, 0.0f), sf::max(d.z, 0.0f)};
    return sf::min(sf::max(d.x, sf::max(d.y, d.z)), 0.0f) + sf::sqrt(dot(m, m));
}
SF_HD float raymarch_scene(vec3 origin) {                                                       // :9-32
    float sdf = 2.0f*100.0f;
    for (int i = 2; i < 8; i++) {
        const float side = (float)(i - 1);
        sdf = sf::min(sdf, sd_box(origin, vec3{0.0f, 0.0f, (float)i}, vec3{side, side, side}));
    }
    return sdf;
}
SF_HD vec4 frag_raymarch(const Frag& f) {
    Camera cam = get_camera(f);
    vec3 delta = cam.target - cam.origin;
    vec3 forward = delta/sf::sqrt(dot(delta, delta));                                          // normalize :41
    float traveled = 0.0f, walk = 0.0f;
    int steps;
    for (steps = 0; steps < 100; steps++) {                                                    // :48-53
        vec3 point = cam.origin + (forward*traveled);
        walk = raymarch_scene(point);
        traveled += walk;
        if (walk < 0.001f || walk > 100.0f) break;
    }
    const float v = 1.0f - sf::sqrt((float)steps)*0.1f;                                        // :58
    return {v, v, v, 1.0f};
}

// ---- mandelbrot.frag, tetration.frag ---------------------------------------------------------------------
SF_HD vec3 palette_magma(float t) {                                                             // shaderflow.glsl:220-224
    return palette(t, vec3{0.01060815f, 0.01808215f, 0.10018654f}, vec3{0.38092887f, 0.12061482f, 0.32506528f},
                   vec3{0.79650140f, 0.10506637f, 0.31063031f}, vec3{0.95922872f, 0.53307513f, 0.37488950f});
}
SF_HD vec4 frag_mandelbrot(const Frag& f) {
    Camera cam = get_camera(f);
    vec3 c3;
    if (cam.out_of_bounds) {
        c3 = palette_magma(0.0f);
    } else {
        vec2 z = cam.gluv - vec2{0.5f, 0.0f};
        const vec2 c = z;
        const int quality = sf::to_int(1000.0f*f.u->iQuality);
        int iter = 0;
        for (; iter < quality; iter++) {
            // length(z) > 3.0: sqrt is monotone and correctly rounded, and RN(sqrt(nextafter(9))) is already above 3, so the test is
            // `z.x*z.x + z.y*z.y > 9` bit for bit (NaN: false either way) — without a square-root sequence per iteration
            if (z.x*z.x + z.y*z.y > 9.0f) break;
            z = vec2{z.x*z.x - z.y*z.y, z.x*z.y + z.y*z.x} + c;                                // cmul(z, z) + c :2-7,21
        }
        c3 = palette_magma(sf::pow(1.0f - (float)iter/(float)quality, 20.0f));                 // :25
    }
    return {c3.x, c3.y, c3.z, 1.0f};
}
struct ComplexNumber { float x, y, r, t; };                                                     // tetration.frag:2-5
SF_HD ComplexNumber complex_power(ComplexNumber a, ComplexNumber b) {                           // :20-25
    ComplexNumber z;
    z.r = sf::pow(a.r, b.x)*sf::exp(-b.y*a.t);
    z.t = b.y*sf::log(a.r) + (b.x*a.t);
    z.x = z.r*sf::cos(z.t);
    z.y = z.r*sf::sin(z.t);
    return z;
}
SF_HD vec4 frag_tetration(const Frag& f) {
    Camera cam = get_camera(f);
    ComplexNumber C;
    C.x = cam.gluv.x; C.y = cam.gluv.y;
    C.r = sf::sqrt(C.x*C.x + C.y*C.y); C.t = sf::atan(C.y, C.x);                               // UpdatePolar :7-11
    ComplexNumber Z = C;
    const int max_steps = 67;
    int it = 0;
    for (it = 0; it < max_steps; it++) {
        Z = complex_power(C, Z);
        if (Z.r > 100.0f) break;
    }
    const float k = (float)(it/max_steps);                                                     // integer division :49
    const float theta = atan2(Z.y, Z.x)/TAU;                                                   // atan2n :394-396
    vec3 c = hsv2rgb(theta, 1.0f, k);
    return {c.x, c.y, c.z, 1.0f};
}

template <int FRAGMENT> SF_HD vec4 shade(const Frag& f) {
    if constexpr (FRAGMENT == FRAG_DEFAULT) return frag_default(f);
    else if constexpr (FRAGMENT == FRAG_VISUALIZER) return frag_visualizer(f);
    else if constexpr (FRAGMENT == FRAG_BARS) return frag_bars(f);
    else if constexpr (FRAGMENT == FRAG_WAVEFORM) return frag_waveform(f);
    else if constexpr (FRAGMENT == FRAG_MULTI_CHILD) return frag_multi_child(f);
    else if constexpr (FRAGMENT == FRAG_MULTI_MAIN) return frag_multi_main(f);
    else if constexpr (FRAGMENT == FRAG_SHADERTOY) return frag_shadertoy(f);
    else if constexpr (FRAGMENT == FRAG_DYNAMICS) return frag_dynamics(f);
    else if constexpr (FRAGMENT == FRAG_AUDIO) return frag_audio(f);
    else if constexpr (FRAGMENT == FRAG_MULTIPASS) return frag_multipass(f);
    else if constexpr (FRAGMENT == FRAG_MOTIONBLUR) return frag_motionblur(f);
    else if constexpr (FRAGMENT == FRAG_LIFE_SIMULATION) return frag_life_simulation(f);
    else if constexpr (FRAGMENT == FRAG_LIFE_VISUALS) return frag_life_visuals(f);
    else if constexpr (FRAGMENT == FRAG_VIDEO) return frag_video(f);
    else if constexpr (FRAGMENT == FRAG_RAYMARCH) return frag_raymarch(f);
    else if constexpr (FRAGMENT == FRAG_MANDELBROT) return frag_mandelbrot(f);
    else if constexpr (FRAGMENT == FRAG_TETRATION) return frag_tetration(f);
    else return frag_missing(f);
}

}  // namespace sf
