// launch_geometry.hpp — host-side bounds of the background window a block of samples needs (shared by capi.hip's choice between the
// visualizer's kernel families and the strip unit's choice between its tiles)
#pragma once

#include "host_state.hpp"
#include "visualizer_kernels.hpp"

namespace sfl {
using namespace sf;

// Upper bound of the background window (in texel cells) that a block of `sx` x `sy` shaded samples needs for
// visualizer.frag's taps — the numbers VisualizerShader::setup derives per block, bounded over the frames of a launch.
// Texels per sample are zoom^2 * background.height / shaded_height on both axes (visualizer.frag:17, gtexture); on the
// tape path the per-frame zoom and blur radius live on the device, so their largest values are used (z <= 0.93,
// intensity <= 0.003). A zoomed / panned camera scales each axis by its slope; any other camera (rolled, tilted) mixes the axes:
// camera_slopes() bounds the four partial derivatives of iCamera.gluv over the screen. The kernel still checks every block
// (a window that does not fit falls back to the generic taps), so the bound decides speed, never results.

// Bounds of |d iCamera.gluv / d gluv| over the screen for a camera that is neither the identity nor axis aligned, from get_camera on
// a 17 x 17 lattice of fragments. A camera rolled about its untouched forward axis (right.z == up.z == 0, forward == z) keeps `t` of
// CameraRay2D constant: the map is affine and the lattice differences ARE the slopes; a tilted camera is projective, where the
// slope inside a lattice cell can exceed the difference across it — half as much again covers it for the tilts a plane in front of
// the camera allows before the horizon enters the screen (and the horizon itself fails the finiteness check).
static bool camera_slopes(const RenderArgs& a, float (&slope)[2][2]) {
    const Uniforms& u = a.u;
    const bool affine = u.iCameraProjection == 0 && u.iCameraRight[2] == 0.0f && u.iCameraUpward[2] == 0.0f
        && u.iCameraForward[0] == 0.0f && u.iCameraForward[1] == 0.0f && u.iCameraForward[2] == 1.0f;
    if (u.iCameraProjection != 0) return false;          // stereoscopic jumps at the centre column, equirectangular wraps
    constexpr int G = 16;
    vec2 hit[G + 1][G + 1];
    for (int j = 0; j <= G; j++) for (int i = 0; i <= G; i++) {
        Frag f{};
        f.u = &u; f.aspect = a.aspect;
        f.gluv = vec2{a.aspect*(2.0f*(float)i/(float)G - 1.0f), 2.0f*(float)j/(float)G - 1.0f};
        f.agluv = f.gluv/vec2{a.aspect, 1.0f};
        const Camera c = get_camera(f);
        if (!(fabsf(c.gluv.x) < 1e6f) || !(fabsf(c.gluv.y) < 1e6f)) return false;
        hit[j][i] = c.gluv;
    }
    const float step_x = 2.0f*a.aspect/(float)G, step_y = 2.0f/(float)G, margin = affine ? 1.002f : 1.5f;
    slope[0][0] = slope[0][1] = slope[1][0] = slope[1][1] = 0.0f;
    for (int j = 0; j <= G; j++) for (int i = 0; i <= G; i++) {
        if (i < G) {
            slope[0][0] = fmaxf(slope[0][0], fabsf(hit[j][i + 1].x - hit[j][i].x)/step_x);
            slope[1][0] = fmaxf(slope[1][0], fabsf(hit[j][i + 1].y - hit[j][i].y)/step_x);
        }
        if (j < G) {
            slope[0][1] = fmaxf(slope[0][1], fabsf(hit[j + 1][i].x - hit[j][i].x)/step_y);
            slope[1][1] = fmaxf(slope[1][1], fabsf(hit[j + 1][i].y - hit[j][i].y)/step_y);
        }
    }
    for (auto& row : slope) for (float& v : row) v *= margin;
    return true;
}

static void visualizer_window_bound(const RenderArgs& a, int sx, int sy, int& tw, int& th) {
    const Tex& bg = a.tex[TEX_BACKGROUND];
    const float zoom2 = a.has_vis ? a.vis.zoom2 : 0.93f*0.93f;
    const float intensity = a.has_vis ? fabsf(a.vis.intensity) : 0.003f;
    const float density = zoom2*(float)bg.height/(float)a.hr;
    // texels along x / y per sample step along x / y
    float xx = density, xy = 0.0f, yx = 0.0f, yy = density;
    if (a.axis_camera && !a.identity_camera) {
        // a zoomed / panned camera: iCamera.gluv is an affine function of gluv per axis (glsl.hpp camera_along_axis): its slopes
        bool behind = false;
        const float sx_ = (camera_along_axis<0>(a.u, 1.0f, a.aspect, behind) - camera_along_axis<0>(a.u, -1.0f, a.aspect, behind))/2.0f;
        const float sy_ = (camera_along_axis<1>(a.u, 1.0f, a.aspect, behind) - camera_along_axis<1>(a.u, -1.0f, a.aspect, behind))/2.0f;
        xx *= fabsf(sx_)*1.001f; yy *= fabsf(sy_)*1.001f;
        if (!(xx == xx) || !(yy == yy)) { xx = yy = 1e9f; }
    } else if (!a.identity_camera) {
        float slope[2][2];
        if (camera_slopes(a, slope)) { xx = density*slope[0][0]; xy = density*slope[0][1]; yx = density*slope[1][0]; yy = density*slope[1][1]; }
        else xx = yy = 1e9f;
    }
    const float rx = intensity*a.bg_scale_x*(float)bg.width*1.101f + 0.001f, ry = intensity*(float)bg.height*1.101f + 0.001f;
    // the affine cameras' blocks bound their window from four corners through the HOST's map and widen it by a slack for the difference
    // to the samples' own chain (VisualizerShader::setup 1a': 0.05 texel + 1e-5 of the four coordinates' magnitudes): the tile holds it
    // for coordinates within one repeat of the background (blocks further out take the generic taps)
    const float slack = a.affine_camera ? 2.0f*(0.05f + 1.0e-5f*4.0f*2.0f*(float)(bg.width > bg.height ? bg.width : bg.height)) : 0.0f;
    tw = (int)floorf(fminf((float)(sx - 1)*xx + (float)(sy - 1)*xy + 2.0f*rx + slack, 1e6f)) + 2;
    th = (int)floorf(fminf((float)(sx - 1)*yx + (float)(sy - 1)*yy + 2.0f*ry + slack, 1e6f)) + 2;
}
static const size_t VIS_LDS_LIMIT = 150*1024;                         // leave room for the static shared state of the kernels

}  // namespace sfl
