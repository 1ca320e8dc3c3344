/*
 * ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Plain-C restatement of the reference's per-frame STFT → spectrogram → fragment → SSAA → read-out
 * path (BrokenSource/ShaderFlow v0.11.3). Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker.
 *
 * Pinning status
 *   audio half (sfo_audio.c): PINNED against tests/golden/ fixtures, captured from the reference's own
 *       numpy code by tests/golden/make_golden.py (tests/test_oracle_audio.py).
 *   pixel half (sfo_pixel.c): PINNED against frames of THE REFERENCE ITSELF, run in the build container: /root/reference's own
 *       Python (ShaderScene.main and everything under it, unmodified) rendering through Mesa llvmpipe (OpenGL 4.5 core, the
 *       software rasteriser BASELINE.json names) with its GLSL as shader.py:190-239 assembles it — tests/golden/mesa.npz and
 *       mesa_4k.npz, written by tests/golden/make_golden_mesa*.py (refhost.py, mesa_shim.c say how): probes of every fragment on
 *       the parity tests' inputs, fifteen example scenes exported by scene.main(), two whole 3840x2160 2xSSAA frames.
 *       tests/test_oracle_mesa.py: byte for byte on several, within 1 LSB elsewhere except where measured and bounded there
 *       (llvmpipe's 8-bit fixed-point filter: up to 1.3 % of the values 2 LSB off where an 8-bit texture is filtered twice at 1:1 —
 *       DEMONSTRATED since round 4: filter.npz holds llvmpipe's filtered values unrounded, sfo_set_llvmpipe_filter reproduces them
 *       bit for bit and every such comparison then meets max <= 1; one supersample across a bar's edge in 1e5 pixels at 4K, listed by
 *       coordinates in mesa_4k_outliers.npz; tetration's chaotic boundary; default.glsl's ring).
 *       Second witness (rounds 1-2): the same GLSL adapted mechanically to GLSL ES 3.00 on Google SwiftShader — gles.npz,
 *       tests/test_oracle_gles.py — which also pins fragment/missing.glsl (it reads an uninitialised output: Mesa's image is
 *       undefined, zero-initialising drivers draw the checkerboard) and texelFetch outside the texture (undefined in GL:
 *       SwiftShader clamps, this oracle reads zero like robust-access desktop drivers).
 */
#ifndef SFO_H
#define SFO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------- */
/* Frame clock and PCM chunking */

/* scheduler.py:87-89,134-173 (freewheel) + scene.py:475-479. Outputs are the values the modules
 * SEE while frame k is produced: time_used[k], dt_used[k], rdt_used[k]. */
void sfo_clock(double fps, double speed, int frames, double* time_used, double* dt_used, double* rdt_used);

/* ffmpeg.py:1281-1333 driven by audio/module.py:447-453: samples appended on frame k and the
 * running `tell`; total_samples bounds the file (reader runs dry → 0). */
void sfo_reader(const double* rdt_used, int frames, int samplerate, int channels,
                int64_t total_samples, int32_t* lengths, int64_t* tell);

/* ---------------------------------------------------------------------------------------------- */
/* STFT */

enum { SFO_WINDOW_HANNING = 0, SFO_WINDOW_HANN_POISSON = 1, SFO_WINDOW_NONE = 2 };
enum { SFO_SCALE_OCTAVE = 0, SFO_SCALE_MEL = 1 };
enum { SFO_INTERP_EULER = 0, SFO_INTERP_DIRAC = 1, SFO_INTERP_SINC = 2 };
enum { SFO_REDUCER_AVERAGE = 0, SFO_REDUCER_RMS = 1, SFO_REDUCER_STD = 2 };

/* spectrogram.py:90-108 */
void sfo_window(int kind, int n, double* out);

/* audio/module.py:137-138 + spectrogram.py:155-171,25-26. pcm is planar (channels, total) float32 of
 * the whole stream; the ring-buffer window of the reference is stream[tell-n-1 : tell-1] with zeros
 * before the start. out: (channels, n/2+1) float32 power. */
void sfo_fft_power(const float* pcm, int64_t total, int channels, int64_t tell,
                   int fft_n, int window_kind, float* out);
void sfo_fft_amplitude(const float* pcm, int64_t total, int channels, int64_t tell,
                       int fft_n, int window_kind, float* out);

/* spectrogram.py:167 samplerate.resample(x, ratio, 'linear') — libsamplerate's linear converter under src_simple; PARITY UNPINNED (the
 * package is neither vendored nor importable here: the published algorithm of src_linear.c restated, see sfo_audio.c). Returns the
 * number of frames generated (<= n_out). */
int sfo_resample_linear(const float* in, int n_in, double ratio, float* out, int n_out);

/* spectrogram.py:186-224 (+ scales :73-87, kernels :44-70). Returns nnz (or -needed if cap is too
 * small). CSR of the (bins, fft_bins) float32 matrix. */
int sfo_filterbank(int scale, int interp, double fmin, double fmax, int bins, int fft_n,
                   double samplerate, int32_t* indptr, int32_t* indices, float* data, int cap);

/* spectrogram.py:175-176: M.dot(P.T) — out is the (bins, channels) C-ordered buffer whose bytes the
 * reference re-views as (2, bins) (spectrogram.py:306) and uploads as RG texels. */
void sfo_csr_dot(const int32_t* indptr, const int32_t* indices, const float* data, int bins,
                 const float* power, int channels, int fft_bins, float* out);

/* piano/notes.py:58-59,74-75 and spectrogram.py:226-245 */
int sfo_note_of_frequency(double frequency, double tuning);
double sfo_frequency_of_note(int note, double tuning);
void sfo_from_notes(int start_note, int end_note, int piano, int bins_in, double tuning,
                    double* fmin, double* fmax, int* bins);

/* ---------------------------------------------------------------------------------------------- */
/* DynamicNumber (dynamics.py:77-255) */

typedef struct {
    double frequency, zeta, response, precision;
    int integrate;
} sfo_dyn_params;

/* coefficient selection dynamics.py:231-242; returns 0 = clamped-k2 branch, 1 = pole matching */
int sfo_dyn_coeffs(const sfo_dyn_params* p, double dt, double* k1, double* k2, double* k3);

/* float32 array system (spectrogram.py:287-290). `previous` holds the last non-early-out target. */
void sfo_dyn_step_f32(const sfo_dyn_params* p, int n, float* value, float* derivative,
                      float* previous, float* integral, const float* target, double dt);

/* float64 scalar system (audio/module.py:413-421, camera.py:147-185) */
void sfo_dyn_step_f64(const sfo_dyn_params* p, double* value, double* derivative, double* previous,
                      double* integral, double target, double dt);

/* ---------------------------------------------------------------------------------------------- */
/* Waveform and loudness */

/* waveform.py:80-87,14-22. out: (points, channels) float32 */
void sfo_waveform_row(const float* pcm, int64_t total, int channels, int64_t tell,
                      int chunk_size, int points, int reducer, float* out);

/* audio/module.py:74-75,457-458 over stream[tell-n-1 : tell-1] of all channels jointly */
void sfo_volume_std(const float* pcm, int64_t total, int channels, int64_t tell, int n,
                    float* volume_target, float* std_target);

/* ---------------------------------------------------------------------------------------------- */
/* Pixel half */

enum { SFO_U8 = 0, SFO_F32 = 1, SFO_U16 = 2, SFO_F16 = 3 };
/* SFO_*_MIPMAP: the minification filter of a texture with mipmaps=True (moderngl_filter, texture.py:131-137); magnification stays
 * LINEAR / NEAREST (OpenGL refuses a mipmap filter there: texture.py:279 hands it the same enum, INVALID_ENUM, no effect) */
enum { SFO_NEAREST = 0, SFO_LINEAR = 1, SFO_LINEAR_MIPMAP = 2, SFO_NEAREST_MIPMAP = 3 };

typedef struct {
    const void* data;      /* row 0 = bottom row (GL order), tightly packed */
    int32_t width, height, components, dtype, filter, repeat_x, repeat_y;
    int32_t levels;        /* 0/1: no chain. Else `mips` holds levels 1 … levels-1 (sfo_mip_offset) */
    const void* mips;
} sfo_texture;

/* texture.build_mipmaps() (texture.py:277-278) as the OpenGL implementation behind the goldens does it — Mesa renders level k+1 as a
 * LINEAR-filtered, edge-clamped blit of level k at the new level's pixel centres (the 2x2 box mean where an extent halves exactly;
 * measured: tests/golden/mip.npz holds llvmpipe's levels). Float weights, unorm results rounded to nearest; under
 * sfo_set_llvmpipe_filter the unorm8 levels come from its fixed-point filter instead (bit-identical to llvmpipe's).
 * sfo_mip_offset(level >= 1): byte offset of that level inside `mips`; level == levels: the size of the whole chain. */
int sfo_mip_levels(int width, int height);
int64_t sfo_mip_offset(const sfo_texture* t, int level);
void sfo_build_mipmaps(const sfo_texture* t, void* mips);

/* Every uniform the in-scope fragments can read (scene.py:687-703, camera.py:196-201 + its nine
 * ShaderDynamics camera.py:147-185, audio/module.py:413-421, spectrogram.py:313-320, waveform.py:89-90) */
typedef struct {
    float iTime, iTau, iDuration, iDeltatime;
    float iResolution[2];
    float iWantAspect, iQuality, iSSAA, iFramerate;
    int32_t iFrame, iRealtime, iLayer, iSubsample;
    float iMouse[2];
    int32_t iMouseInside, iMouse1, iMouse2;
    int32_t iCameraMode, iCameraProjection;
    float iCameraRight[3], iCameraUpward[3], iCameraForward[3];
    float iCameraPosition[3], iCameraZenith[3];
    float iCameraSeparation, iCameraZoom, iCameraIsometric, iCameraFocalLength, iCameraOrbital, iCameraDolly;
    float iAudioVolume, iAudioVolumeIntegral, iAudioSTD;
    int32_t iSpectrogramLength, iSpectrogramBins, iSpectrogramSmooth, iSpectrogramScroll;
    float iSpectrogramOffset, iSpectrogramMin, iSpectrogramMax;
    int32_t iWaveformLength;
    float user[16];        /* scene-defined float uniforms, by slot */
} sfo_uniforms;

enum {
    SFO_FRAG_DEFAULT = 0,      /* resources/shaders/fragment/default.glsl   */
    SFO_FRAG_MISSING = 1,      /* resources/shaders/fragment/missing.glsl   */
    SFO_FRAG_VISUALIZER = 2,   /* examples/basic/shaders/visualizer.frag    */
    SFO_FRAG_BARS = 3,         /* examples/basic/shaders/bars.frag          */
    SFO_FRAG_WAVEFORM = 4,     /* examples/basic/shaders/waveform.frag      */
    SFO_FRAG_MULTI_CHILD = 5,  /* examples/basic/demo.py:74-79 (inline)     */
    SFO_FRAG_MULTI_MAIN = 6,   /* examples/basic/demo.py:83-89 (inline)     */
    SFO_FRAG_SHADERTOY = 7,    /* examples/basic/shaders/shadertoy.frag     */
    SFO_FRAG_DYNAMICS = 8,     /* examples/basic/demo.py:121-126 (inline)   */
    SFO_FRAG_AUDIO = 9,        /* examples/basic/demo.py:149-153 (inline)   */
    SFO_FRAG_MULTIPASS = 10,   /* examples/basic/shaders/multipass.frag     */
    SFO_FRAG_MOTIONBLUR = 11,  /* examples/basic/shaders/motionblur.frag    */
    SFO_FRAG_LIFE_SIMULATION = 12, /* examples/basic/shaders/life/simulation.glsl */
    SFO_FRAG_LIFE_VISUALS = 13,    /* examples/basic/shaders/life/visuals.glsl    */
    SFO_FRAG_VIDEO = 14,       /* examples/basic/shaders/video.frag         */
    SFO_FRAG_RAYMARCH = 15,    /* examples/basic/shaders/raymarch.frag      */
    SFO_FRAG_MANDELBROT = 16,  /* examples/fractals/shaders/mandelbrot.frag */
    SFO_FRAG_TETRATION = 17,   /* examples/fractals/shaders/tetration.frag  */
};

/* slots 4.. are the temporal history `<name>{t}x0` of the texture the fragment reads by coordinates
 * (texture.py:346-347, 380-381): iScreen for multipass/motionblur, iLife for life, iVideo for video */
enum { SFO_TEX_BACKGROUND = 0, SFO_TEX_SPECTROGRAM = 1, SFO_TEX_WAVEFORM = 2, SFO_TEX_CHILD = 3,
       SFO_TEX_HISTORY = 4, SFO_TEX_HISTORY_DEPTH = 12, SFO_TEX_SLOTS = 16 };

/* shader.py:388-405 for one layer: evaluate `fragment` at every pixel centre of a (wr, hr) target
 * and store RGBA8 (rows bottom-up). Only rows [y0, y1) are produced (band rendering for the bounded
 * CPU baseline); `threads` > 1 splits rows over pthreads. out has wr*hr*4 bytes. */
void sfo_render(int fragment, const sfo_uniforms* u, const sfo_texture* textures /*[SFO_TEX_SLOTS]*/,
                int wr, int hr, int y0, int y1, int threads, uint8_t* out);

/* Same, into a target of `components` channels of SFO_U8 or SFO_F32 (texture.py:177-184: any ShaderTexture
 * format can be a render target; life/simulation renders into R32F, demo.py:236-238) */
void sfo_render_to(int fragment, const sfo_uniforms* u, const sfo_texture* textures,
                   int wr, int hr, int y0, int y1, int threads, int components, int dtype, void* out);

/* fragment/final.glsl:1-33 + shader.py:391-396: iScreen (RGBA8, linear, clamp) → RGB8 (w, h),
 * rows [y0, y1). */
void sfo_resolve(const uint8_t* screen, int wr, int hr, int w, int h, int subsample,
                 int y0, int y1, int threads, uint8_t* out);

/* Checker's switch, off by default: filter unorm8 textures as Mesa llvmpipe does (24.8 fixed-point coordinates, 8-bit weights, every
 * lerp rounded back to 8 bits — sfo_pixel.c). It exists to DEMONSTRATE the cause of the > 1 LSB values against the llvmpipe goldens
 * (tests/test_oracle_mesa.py); parity tests of the HIP kernels run with it off. Process-wide; set it outside sfo_render calls. */
void sfo_set_llvmpipe_filter(int on);

/* RGB8 (w x h, even) → planar yuv420p (w*h*3/2 bytes): the product's own definition of the optional device-side conversion
 * (sfx_rgb_to_yuv420), restated; matrix 0 = BT.601 limited, 1 = BT.709 limited */
void sfo_rgb_to_yuv420(const uint8_t* rgb, int w, int h, int matrix, uint8_t* yuv);

/* GL `texture()` on one coordinate, exposed for sampler unit tests */
void sfo_sample(const sfo_texture* t, float s, float tt, float rgba[4]);

/* … with the implicit derivatives of a 2x2 quad: the coordinates of the pixel's horizontal and vertical quad neighbours (mipmapped
 * textures: level of detail per OpenGL 3.3 section 3.8.11; other filters ignore them) */
void sfo_sample_quad(const sfo_texture* t, float s, float tt, float s_right, float t_right, float s_above, float t_above, float rgba[4]);

/* sfmath entry points for the accuracy / cross-implementation tests */
float sfo_test_math(int fn, float a, float b);

#ifdef __cplusplus
}
#endif
#endif
