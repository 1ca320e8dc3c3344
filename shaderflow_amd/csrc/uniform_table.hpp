// uniform_table.hpp — the built-in uniforms by name: where sfx_uniform_set stores what the scene sends (scene.py:687-703,
// camera.py:196-201, audio/module.py:413-421, spectrogram.py:313-320, waveform.py:89-90) and their values before the first
// frame. Shared by the C-ABI (capi.hip) and by the host-side check of translated fragments (jit_runtime.hpp, SF_JIT_HOST).
#pragma once

#include "glsl.hpp"

#include <cstddef>
#include <cstring>

namespace sf {

struct UniformField { const char* name; size_t offset; int count; bool integer; };
#define UF(n, c, i) {#n, offsetof(Uniforms, n), c, i}
static const UniformField g_uniform_fields[] = {
    UF(iTime, 1, false), UF(iTau, 1, false), UF(iDuration, 1, false), UF(iDeltatime, 1, false), UF(iResolution, 2, false),
    UF(iWantAspect, 1, false), UF(iQuality, 1, false), UF(iSSAA, 1, false), UF(iFramerate, 1, false),
    UF(iFrame, 1, true), UF(iRealtime, 1, true), UF(iLayer, 1, true), UF(iSubsample, 1, true),
    UF(iMouse, 2, false), UF(iMouseInside, 1, true), UF(iMouse1, 1, true), UF(iMouse2, 1, true),
    UF(iCameraMode, 1, true), UF(iCameraProjection, 1, true),
    UF(iCameraRight, 3, false), UF(iCameraUpward, 3, false), UF(iCameraForward, 3, false),
    UF(iCameraPosition, 3, false), UF(iCameraZenith, 3, false),
    UF(iCameraSeparation, 1, false), UF(iCameraZoom, 1, false), UF(iCameraIsometric, 1, false),
    UF(iCameraFocalLength, 1, false), UF(iCameraOrbital, 1, false), UF(iCameraDolly, 1, false),
    UF(iAudioVolume, 1, false), UF(iAudioVolumeIntegral, 1, false), UF(iAudioSTD, 1, false),
    UF(iSpectrogramLength, 1, true), UF(iSpectrogramBins, 1, true), UF(iSpectrogramSmooth, 1, true), UF(iSpectrogramScroll, 1, true),
    UF(iSpectrogramOffset, 1, false), UF(iSpectrogramMin, 1, false), UF(iSpectrogramMax, 1, false),
    UF(iWaveformLength, 1, true),
};

static inline void default_uniforms(Uniforms& u) {
    memset(&u, 0, sizeof u);
    u.iResolution[0] = 1920; u.iResolution[1] = 1080; u.iWantAspect = 1920.0f/1080.0f;
    u.iQuality = 0.5f; u.iSSAA = 1.0f; u.iFramerate = 60.0f; u.iDuration = 10.0f; u.iSubsample = 2;
    u.iCameraMode = 1;                                              // Camera2D, camera.py:134
    u.iCameraRight[0] = 1.0f; u.iCameraUpward[1] = 1.0f; u.iCameraForward[2] = 1.0f; u.iCameraZenith[1] = 1.0f;
    u.iCameraSeparation = 0.05f; u.iCameraZoom = 1.0f; u.iCameraFocalLength = 1.0f;   // camera.py:147-185
}

}  // namespace sf
