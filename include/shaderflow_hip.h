/*
 * shaderflow_hip.h — C-ABI of libshaderflow_hip.so: the MI355X (gfx950) implementation of ShaderFlow's
 * per-frame STFT → spectrogram → per-pixel fragment → SSAA resolve → frame read-out path.
 *
 * The reference (BrokenSource/ShaderFlow v0.11.3) has no FFI for this path: it calls numpy/scipy and,
 * through moderngl, an OpenGL driver. Each entry point below replaces one group of those third-party
 * calls; the reference call site it stands for is cited as file:line (paths relative to the reference
 * repository root). The Python host package (shaderflow_amd/) binds these with ctypes; INTEGRATION.md
 * shows the stub a maintainer of the reference would add.
 *
 * Conventions: plain C types only; opaque 64-bit handles; every function returns SFX_OK (0) or a negative
 * SFX_E_* code and leaves a message for sfx_last_error() (thread-local). Host pointers are borrowed for the
 * duration of the call. All work of a context is issued on ONE HIP stream (its own, or the caller's when
 * `stream` is given to sfx_ctx_create) — like the reference, where everything runs on the GL context's
 * thread (scene.py:128-195). Calls on one context must come from one thread at a time.
 * Image rows are bottom-up everywhere (OpenGL order), exactly what `fbo.read` hands to ffmpeg's `vflip`
 * (exporting.py:94-103,170-174).
 */
#ifndef SHADERFLOW_HIP_H
#define SHADERFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t sfx_handle;

enum {
    SFX_OK = 0,
    SFX_E_INVALID = -1,        /* bad argument / handle                                    */
    SFX_E_HIP = -2,            /* a HIP runtime call failed (message carries hipGetErrorString) */
    SFX_E_NO_DEVICE = -3,      /* no gfx950 device visible                                  */
    SFX_E_UNSUPPORTED = -4,    /* combination not implemented by a kernel                   */
    SFX_E_IO = -5,             /* pipe / file descriptor write failed                       */
    SFX_E_TOO_LARGE = -6,      /* texture exceeds the context limit (texture.py:251-252)    */
};

enum { SFX_U8 = 0, SFX_F32 = 1, SFX_U16 = 2, SFX_F16 = 3 };          /* numpy2mgltype, texture.py:28-38 */
enum { SFX_NEAREST = 0, SFX_LINEAR = 1,                               /* TextureFilter, texture.py:42-44 */
       SFX_LINEAR_MIPMAP_LINEAR = 2, SFX_NEAREST_MIPMAP_NEAREST = 3 };      /* … with mipmaps=True: moderngl_filter, texture.py:131-137 */
enum { SFX_WINDOW_HANNING = 0, SFX_WINDOW_HANN_POISSON = 1, SFX_WINDOW_NONE = 2 };   /* spectrogram.py:90-108 */
enum { SFX_REDUCER_AVERAGE = 0, SFX_REDUCER_RMS = 1, SFX_REDUCER_STD = 2 };          /* waveform.py:14-22 */

/* GLSL uniform types accepted by sfx_uniform_set (variable.py:12-23) */
enum { SFX_T_FLOAT = 0, SFX_T_INT = 1, SFX_T_BOOL = 2, SFX_T_VEC2 = 3, SFX_T_VEC3 = 4, SFX_T_VEC4 = 5,
       SFX_T_MAT2 = 6, SFX_T_MAT3 = 7, SFX_T_MAT4 = 8 /* column major, 4/9/16 floats; loaded programs only */ };

const char* sfx_last_error(void);
const char* sfx_version(void);
/* Fingerprint of the kernel-argument layout (member offsets + build switches) the library was built with. Code objects of
 * run-time translated fragments carry the fingerprint of the headers THEY were compiled against; sfx_program_load refuses a
 * mismatch, and hosts key their code-object caches with it. */
uint64_t sfx_abi_layout(void);
/* Name of the kernel instance the calling thread's last sfx_render / sfx_render_resolve / sfx_render_tape launched for the
 * fragment, e.g. "k_render_resolve<VisualizerShader<80, 10, 8, 1, 1, 128, 128, 2>, 2>": tests and profiles tie results to it. */
const char* sfx_last_kernel(void);

/* ------------------------------------------------------------------------------------------------ */
/* Context — replaces window/GL-context creation (scene.py:145-157) and GL_MAX_VIEWPORT_DIMS (texture.py:251) */

typedef struct {
    char device_name[128];
    char gcn_arch[64];
    int32_t device_id;
    int32_t compute_units;
    int32_t max_texture_dim;       /* what ShaderTexture.make checks against */
    int32_t wavefront_size;
    int64_t total_memory;
    int64_t lds_per_cu;
} sfx_ctx_info_t;

int sfx_ctx_create(int device_id, void* stream /* hipStream_t or NULL */, sfx_handle* ctx);
int sfx_ctx_info(sfx_handle ctx, sfx_ctx_info_t* info);
int sfx_ctx_synchronize(sfx_handle ctx);
/* Encoder hand-off (exporting.py:94-103 adds ffmpeg's `vflip` filter because GL rows are bottom-up): with enabled != 0
 * sfx_resolve / sfx_render_resolve / sfx_render_tape write their RGB8 frames top-down, so no filter is needed. */
int sfx_ctx_output_top_down(sfx_handle ctx, int enabled);
/* How LINEAR unorm8 textures of this context are filtered. OpenGL leaves the precision of the bilinear weights to the implementation
 * (3.3 core section 3.8.11: >= 4 subtexel bits), so "the reference's OpenGL path" (shader.py:388-405 drawing through moderngl) is a
 * family: SFX_FILTER_SPEC (default) uses float weights, as GPUs' drivers and SwiftShader do to within their precision; SFX_FILTER_FIXED8
 * is Mesa llvmpipe's unorm8 path — the software rasteriser north_star names as the reference's CPU path: 24.8 fixed-point texel
 * coordinates, 8-bit weights, every lerp rounded back to 8 bits — measured bit for bit on it (tests/golden/filter.npz). With it frames
 * meet llvmpipe's within 1 LSB everywhere (float weights: up to 1.3 % of the values 2 LSB off where a filtered value is filtered or
 * quantised again). Costs the table-driven and fused kernels (every draw becomes the generic kernel + the resolve pass). */
enum { SFX_FILTER_SPEC = 0, SFX_FILTER_FIXED8 = 1 };
int sfx_ctx_filter_model(sfx_handle ctx, int model);
/* The context's two copy streams (read-out ring = turbopipe's role, exporting.py:147-171; shared-memory ring; peer windows) are chosen
 * once, on first use or by this call, so that neither shares a hardware queue with the render stream: HIP folds its streams onto a
 * few in-order queues, and a copy stream on the render stream's queue serialises read-out and render (C3: 2 080 → 1 215 frames/s).
 * Candidates are created until two are found whose copies complete while a kernel holds the render stream. Reports how many
 * streams were looked at and how many of them ran in series with the render stream (either may be NULL). */
int sfx_ctx_copy_streams(sfx_handle ctx, int* candidates, int* colliding);
/* Tuning aid for the LDS-tiled visualizer kernels: the number of blocks, since the previous call, whose tap window did not fit the
 * tile chosen for the launch and which therefore ran the generic taps (same pixels, about 20 times slower). The first call starts
 * the count and reports 0. No counterpart in the reference. */
int sfx_ctx_tile_misses(sfx_handle ctx, unsigned long long* blocks);
int sfx_ctx_destroy(sfx_handle ctx);

/* Timing on the context's stream with HIP events (bench.py roofline leg). slot in [0, 64). */
int sfx_event_record(sfx_handle ctx, int slot);
int sfx_event_elapsed_ms(sfx_handle ctx, int start_slot, int stop_slot, float* ms);

/* ------------------------------------------------------------------------------------------------ */
/* Textures — replace opengl.texture/framebuffer and their setters (texture.py:261-283), write
 * (texture.py:313-325), release (texture.py:63-67), fbo.read (scene.py:441) */

int sfx_texture_create(sfx_handle ctx, int width, int height, int components, int dtype, sfx_handle* tex);
int sfx_texture_params(sfx_handle tex, int filter, int repeat_x, int repeat_y);
/* viewport (x, y, w, h) in texels, row 0 = bottom; w == h == 0 means the whole texture */
int sfx_texture_write(sfx_handle tex, const void* data, size_t nbytes, int x, int y, int w, int h);
int sfx_texture_read(sfx_handle tex, void* data, size_t nbytes);
/* texture.build_mipmaps() (texture.py:277-278): levels 1… of the chain from the current level 0 (each level the LINEAR-filtered,
 * edge-clamped half-size image of the one above: the 2x2 box mean where an extent halves exactly). A later sfx_texture_write changes
 * level 0 only, as glTexSubImage2D does. Sampled by the filters SFX_LINEAR_MIPMAP_LINEAR / SFX_NEAREST_MIPMAP_NEAREST
 * (sfx_texture_params), level of detail from the implicit derivatives of the coordinate (OpenGL 3.3 section 3.8.11). */
int sfx_texture_build_mipmaps(sfx_handle tex);
int sfx_texture_read_level(sfx_handle tex, int level, void* data, size_t nbytes);
int sfx_texture_device_ptr(sfx_handle tex, void** ptr, size_t* nbytes);
int sfx_texture_destroy(sfx_handle tex);

/* ------------------------------------------------------------------------------------------------ */
/* Programs — replace opengl.program(vs, fs) (shader.py:324), program[name].value = v (shader.py:353-360),
 * sampler binding (shader.py:377-386) and fbo.use(); vao.render() (shader.py:367-375, 388-405).
 *
 * The library itself has no GLSL compiler: `source` (the USER part of the fragment, shader.py:229-235) is normalised
 * (comments and whitespace stripped) and looked up in the registry of fragments restated as HIP kernels.
 * `*fallback` is set to 1 when it is unknown and the `missing.glsl` kernel was bound instead — the
 * reference's compile-error behaviour (shader.py:323-340). A registry name ("visualizer", "default" …)
 * is accepted in place of the source. */

int sfx_program_lookup(sfx_handle ctx, const char* source, sfx_handle* program, int* fallback);

/* Fragments outside the registry — the other half of opengl.program(vs, fs) (shader.py:313-349). The host translates the
 * GLSL to HIP C++ and compiles it to a gfx950 code object (shaderflow_amd/glsl2hip.py; hipcc --genco against
 * csrc/jit_runtime.hpp); this call loads it. `bindings` name what the host may set on it: a uniform lives at floats
 * [slot, slot + count) of the program's scene-defined uniform block (integer: stored as int32 bits), a sampler at texture
 * slot `slot`. The program then behaves like any other (sfx_uniform_set, sfx_sampler_bind, sfx_render, sfx_render_resolve,
 * sfx_render_tape). Fails with SFX_E_INVALID if the code object was compiled against other kernel headers than this library. */
typedef struct sfx_binding {
    const char* name;
    int sampler;        /* 1: sampler2D, 0: uniform */
    int slot;
    int count;          /* uniform: number of 32-bit values (1-4); sampler: 1 */
    int integer;        /* uniform: int/bool/uint rather than float */
} sfx_binding;
int sfx_program_load(sfx_handle ctx, const void* code_object, size_t nbytes, const sfx_binding* bindings, int nbindings, sfx_handle* program);
const char* sfx_program_name(sfx_handle program);
/* 1 when sfx_render_resolve / the fused branch of sfx_render_tape can run the program at this SSAA factor, 0 when it needs
 * sfx_render + sfx_resolve (a translated fragment that calls dFdx/dFdy/fwidth needs its 2x2 neighbours in the lanes of a quad:
 * the unfused kernel lays them out so, the fused kernel does for ssaa == 2 only) */
int sfx_program_fusable(sfx_handle program, int ssaa);
/* returns SFX_OK whether or not the program reads `name` (inactive uniforms are ignored, shader.py:356-357);
 * *known (may be NULL) tells which */
int sfx_uniform_set(sfx_handle program, const char* name, int type, const void* value, int* known);
int sfx_sampler_bind(sfx_handle program, const char* name, sfx_handle tex, int* known);
/* … `count` of them in one call (the samplers of a temporal x layers matrix after texture.roll(), texture.py:295-298, 351-381) */
int sfx_sampler_bind_many(sfx_handle program, const char* const* names, const sfx_handle* textures, int count);
/* iTime, iTau, iDeltatime, iFrame (scene.py:687-703): what a frame changes when only the clock moves, in one call */
int sfx_uniform_set_clock(sfx_handle program, float time, float tau, float deltatime, int frame);
int sfx_program_destroy(sfx_handle program);

/* One draw of the fullscreen quad into `target` (any format; RGBA8 for iScreen): shader.py:401-403 */
int sfx_render(sfx_handle program, sfx_handle target, int layer);
/* fragment/final.glsl as a pass: src (RGBA8 iScreen) → dst (RGB8 iFinal): shader.py:391-396 */
int sfx_resolve(sfx_handle ctx, sfx_handle src, sfx_handle dst, int subsample);
/* Both fused: shade ssaa x ssaa supersamples per output pixel, quantise to the RGBA8 value iScreen would
 * hold, resolve in registers, write RGB8 `final` only. Returns SFX_E_UNSUPPORTED for (ssaa, subsample)
 * pairs whose final.glsl footprint leaves the pixel's own block — use the two calls above then. */
int sfx_render_resolve(sfx_handle program, sfx_handle final_tex, int ssaa, int subsample);
int sfx_fused_supported(int ssaa_numerator_x1000, int subsample);

/* ------------------------------------------------------------------------------------------------ */
/* Frame read-out — replaces make_buffers / fbo.read_into / turbopipe.pipe|sync|done (exporting.py:140-174) */

int sfx_ring_create(sfx_handle ctx, size_t frame_bytes, int slots, sfx_handle* ring);
int sfx_ring_read_async(sfx_handle ring, sfx_handle tex, int slot);                 /* fbo.read_into(buffer) */
int sfx_ring_read_device_async(sfx_handle ring, const void* device_ptr, int slot);  /* same, from a raw frame */
/* Batched producers (the frame tape renders many frames per launch): mark "everything launched so far is what the
 * next reads may depend on" with one of two fences, read frames against that fence, and make the render stream
 * wait for a slot's copy before the frame buffer it came from is overwritten. */
int sfx_ring_fence(sfx_handle ring, int which /* 0 or 1 */);
int sfx_ring_read_fenced_async(sfx_handle ring, const void* device_ptr, int slot, int which);
int sfx_ring_stream_wait(sfx_handle ring, int slot);
/* `count` consecutive frames of a batch (first at device_ptr, `stride` bytes apart) read against fence `which` (< 0: against the
 * render stream as it is now) into slots first_slot, first_slot + 1, … (modulo the ring) and queued for `fd` in that order:
 * the loop of read_into + turbopipe.pipe over a batch (exporting.py:151-174) in one call. */
int sfx_ring_pipe_frames(sfx_handle ring, const void* device_ptr, size_t stride, int count, int first_slot, int which, int fd);
int sfx_ring_sync(sfx_handle ring, int slot, void** host_ptr);                      /* buffer.read() */
int sfx_ring_pipe(sfx_handle ring, int slot, int fd);                               /* turbopipe.pipe */
int sfx_ring_pipe_sync(sfx_handle ring, int slot);                                  /* turbopipe.sync; slot < 0: all */
int sfx_ring_destroy(sfx_handle ring);

/* Encoder hand-off, optional half (exporting.py:94-134): `frames` RGB8 frames (consecutive, width*height*3 bytes each) on
 * the device → planar yuv420p (I420: Y, U, V; width*height*3/2 bytes each) on the device, on the context's stream. BT.601 limited
 * range (matrix 0) or BT.709 limited (1) in 8-bit integer arithmetic, chroma from the rounded 2x2 mean of R, G, B — defined in
 * capi_readout.hip, restated in the oracle. Half the bytes over PCIe and the pipe: ffmpeg takes `-pix_fmt yuv420p` rawvideo as it is. */
int sfx_rgb_to_yuv420(sfx_handle ctx, const void* rgb, void* yuv, int width, int height, int frames, int matrix);

/* The frame loop of a scene in which nothing but the clock moves (layered / temporal scenes without host logic: demo.py's Multipass,
 * MotionBlur, Life), `nframes` frames in ONE call: scene.next (scene.py:456-479) = every program's render (shader.py:388-405: a draw per
 * layer into row 0 of its texture matrix, then texture.roll(), texture.py:295-298), iFinal's resolve (shader.py:391-396), exporting.pipe
 * (exporting.py:151-174). `passes` in the order scene.next renders; `matrices[m].textures` is [temporal][layers] in the matrix' CURRENT
 * order (the call rolls its own copy: the host rolls its matrices by `nframes` afterwards), `names` the sampler name of every box or
 * NULL. `clock[f]` = iTime, iTau, iDeltatime, iFrame of frame f. With a ring and fd >= 0 every frame is read out and piped
 * (slot (first_slot + f) % slots; `planar_slots`: device staging per slot for yuv420p, or NULL for rgb24). */
enum { SFX_PASS_LAYERS = 0, SFX_PASS_FUSED = 1, SFX_PASS_RESOLVE = 2 };
typedef struct sfx_sequence_pass { sfx_handle program; int kind; int matrix; sfx_handle target; int ssaa; int subsample; } sfx_sequence_pass;
typedef struct sfx_sequence_matrix { int temporal, layers; const sfx_handle* textures; const char* const* names; } sfx_sequence_matrix;
typedef struct sfx_clock_tick { float time, tau, deltatime; int32_t frame; } sfx_clock_tick;      /* sfx_uniform_set_clock's arguments */
int sfx_clock_sequence_run(sfx_handle ctx, const sfx_sequence_pass* passes, int npasses, const sfx_sequence_matrix* matrices, int nmatrices,
                           const sfx_clock_tick* clock, int nframes, sfx_handle ring, int first_slot, int fd,
                           void* const* planar_slots, int yuv_matrix, int width, int height);

/* ------------------------------------------------------------------------------------------------ */
/* Cross-process frame queue of a sharded export (one process per GPU; no reference equivalent). The sink takes
 * one byte stream, so one process owns it (rank 0) — but every rank reads its finished frames out over its OWN PCIe link into a
 * POSIX shared-memory segment (`name` must start with '/'), and a writer thread in rank 0 hands them to the file descriptor in
 * the order given to sfx_shm_drain. Per rank a ring of `slots` frames; sfx_shm_push blocks while the rank's ring is full.
 * What turbopipe.pipe/sync (exporting.py:147-171) is to one process, this is to N. */
int sfx_shm_create(sfx_handle ctx, const char* name, int rank, int world, size_t frame_bytes, int slots, sfx_handle* shm);
/* The rank's next frame (device memory, complete on the context's stream when this call is made): asynchronous copy, published
 * to the writer when it has landed. */
int sfx_shm_push(sfx_handle shm, const void* device_ptr);
/* Every frame pushed so far has left its device buffer and is visible to the writer. */
int sfx_shm_flush(sfx_handle shm);
/* The first `frames` frames this rank pushed have left their device buffers (a batch buffer may be rendered into again). */
int sfx_shm_wait(sfx_handle shm, int64_t frames);
/* Rank 0: start the writer. Frames are written to `fd` run by run: counts[k] frames of rank ranks[k], each rank's frames in the
 * order that rank pushed them. fd < 0: no sink — the frames are consumed in the same order and discarded. */
int sfx_shm_drain(sfx_handle shm, int fd, const int32_t* ranks, const int32_t* counts, int runs);
int sfx_shm_drain_wait(sfx_handle shm);      /* the writer has written every run (or failed: SFX_E_IO) */
int sfx_shm_abort(sfx_handle shm);           /* any rank: every process of the group stops waiting (its calls fail with SFX_E_IO) */
int sfx_shm_unlink(sfx_handle shm);          /* rank 0, after every rank has mapped the segment: its name goes, a crash cannot leak it */
int sfx_shm_destroy(sfx_handle shm);         /* unmaps; rank 0 also unlinks the segment if sfx_shm_unlink did not */

/* ------------------------------------------------------------------------------------------------ */
/* Audio — replaces BrokenAudio's ring (audio/module.py:113-138), np.hanning/np.fft.rfft/csr.dot
 * (spectrogram.py:103,170,176), the waveform reduce (waveform.py:80-87) and RMS/std (audio/module.py:457-458).
 * In export mode the whole file is known, so the "ring" is the PCM itself, resident in HBM: the window
 * the reference reads after `tell` samples were appended is stream[tell-n-1 : tell-1], zeros before 0. */

/* PCM ingest without an ffmpeg binary (the reference pipes every file through ffmpeg, ffmpeg.py:1194-1235,
 * 1240-1333): FLAC streams decoded natively — host code, integer-exact by the format's definition, CRC-checked. `data` is the
 * whole file. samples = per channel; out = interleaved float32 (integer / 2^(bits-1), like ffmpeg's pcm_f32le). */
int sfx_flac_info(const void* data, size_t nbytes, int64_t* samples, int* channels, int* samplerate, int* bits);
int sfx_flac_decode(const void* data, size_t nbytes, float* out, int64_t capacity_floats, int64_t* samples_written);

int sfx_audio_upload(sfx_handle ctx, const float* interleaved, int64_t samples, int channels,
                     int samplerate, sfx_handle* audio);
int sfx_audio_destroy(sfx_handle audio);

/* The filterbank is built by the host exactly as spectrogram.py:194-224 does and handed over as CSR. */
int sfx_stft_plan(sfx_handle ctx, int fft_n, int window, int bins, int channels,
                  const int32_t* indptr, const int32_t* indices, const float* data, sfx_handle* plan);
/* FourierMagnitude (spectrogram.py:20-26): what `fft()` makes of the complex bins. Power (the default) = (x*conj(x)).real,
 * Amplitude = np.abs(x); both evaluated in float64 and cast to float32 like the reference (:169-171). */
enum { SFX_MAGNITUDE_POWER = 0, SFX_MAGNITUDE_AMPLITUDE = 1 };
/* `sample_rateio != 1` (spectrogram.py:144-167): the transform takes fft_size = int(2**fft_n * ratio) samples (even, <= 16384), sample
 * n being (float)(in[tap_a[n]] + tap_w[n]*(in[tap_b[n]] - in[tap_a[n]])), evaluated in float64, over the last 2**fft_n ring samples:
 * the read positions of samplerate.resample(x, ratio, 'linear') (spectrogram.py:167: libsamplerate's linear converter), which depend
 * on the sizes and the ratio only. The CSR matrix has fft_size/2 + 1 columns; built-in windows are evaluated for fft_size. */
int sfx_stft_plan_resampled(sfx_handle ctx, int fft_n, int fft_size, const int32_t* tap_a, const int32_t* tap_b, const double* tap_w,
                            int window, int bins, int channels, const int32_t* indptr, const int32_t* indices, const float* data, sfx_handle* plan);
int sfx_stft_plan_magnitude(sfx_handle plan, int magnitude);
/* A window function of the caller's own (spectrogram.py:90-108 lets `window` be any callable N -> array): its float64 values
 * replace the plan's table; n = 2**fft_n. */
int sfx_stft_plan_window(sfx_handle plan, const double* window, int n);
int sfx_stft_plan_destroy(sfx_handle plan);

/* Per-frame entry points (the faithful frame loop: results come back to the host like numpy arrays).
 * tell[k] = samples read when frame k is produced. */
int sfx_stft_power(sfx_handle plan, sfx_handle audio, const int64_t* tell, int nframes,
                   float* power /* [nframes][channels][fft_bins] */);
/* np.fft.rfft(window*frame) itself, float64 (re, im) pairs: spectrum[frame][channel][fft_bins][2] — for a `magnitude` callable of the
 * user's own (reference: shaderflow/audio/spectrogram.py:20-41 accepts any callable on the complex spectrum, :169-171 applies it); the host
 * applies the callable and hands its float32 result [frame][channel][fft_bins] to sfx_filterbank_apply (= :175-176 on given magnitudes). */
int sfx_stft_spectrum(sfx_handle plan, sfx_handle audio, const int64_t* tell, int nframes, double* spectrum);
int sfx_filterbank_apply(sfx_handle plan, const float* magnitudes, int nframes, int use_mfma, float* out /* [nframes][bins][channels] */);
/* M.dot(fft().T): out[frame][bin][channel] — the (bins, 2) buffer of spectrogram.py:176,306.
 * use_mfma = 1: dense banded GEMM on v_mfma_f32_32x32x2_f32; 0: CSR rows in scipy's order (bit-exact). */
int sfx_spectrogram_targets(sfx_handle plan, sfx_handle audio, const int64_t* tell, int nframes,
                            int use_mfma, float* out);
int sfx_waveform_rows(sfx_handle audio, const int64_t* tell, int nframes, int chunk_size, int points,
                      int reducer, float* out /* [nframes][points][channels] */);
int sfx_volume_std(sfx_handle audio, const int64_t* tell, int nframes, int window_samples,
                   float* out /* [nframes][2] = volume target, std target */);

/* ------------------------------------------------------------------------------------------------ */
/* Frame tape — the batched export path (no reference equivalent): for a run of frames the
 * device computes everything the modules would have produced frame by frame and keeps it in HBM:
 * the spectrogram column after DynamicNumber smoothing, the waveform row, and the audio uniforms.
 * DynamicNumber (dynamics.py:197-250) coefficients depend only on dt and are computed by the host. */

typedef struct {
    float dt, k1, k2, k3;          /* python scalars rounded to float32 (numpy NEP 50) */
} sfx_dyn_coeff_f32;
typedef struct {
    double dt, k1, k2, k3;
} sfx_dyn_coeff_f64;
typedef struct {
    float iTime, iTau, iSpectrogramOffset;
    int32_t iFrame;
} sfx_frame_clock;

/* DynamicNumber.next on its own (dynamics.py:197-250; SURVEY.md §8b "dynamics_scan"): the float32 array system of
 * ShaderSpectrogram (spectrogram.py:287-290, 304-307) walked over `nframes` targets. Host arrays: targets and values
 * [nframes][n]; coeff[frame] = the python scalars of that frame (dt == 0: the frame is skipped, :210-211); state =
 * value | derivative | previous (n floats each), read and written back. The early-out `max|target - value| < precision`
 * (:222-225) freezes the whole system for the frame, exactly as the reference does. n <= 2048. */
int sfx_dynamics_scan(sfx_handle ctx, int nframes, int n, const float* targets, const sfx_dyn_coeff_f32* coeff,
                      float precision, float* state, float* values);
/* The same for `nsystems` independent float64 scalar systems (volume, std: audio/module.py:413-421; the camera's:
 * camera.py:147-185). targets [nframes][nsystems], coeff [nsystems][nframes], state [nsystems][4] = value, derivative,
 * previous, integral (in/out), out [nframes][nsystems][3] = value, integral, derivative. */
int sfx_dynamics_scan_f64(sfx_handle ctx, int nframes, int nsystems, const double* targets, const sfx_dyn_coeff_f64* coeff,
                          double precision, int integrate, double* state, double* out);

typedef struct {
    int32_t points, chunk_size, reducer;     /* ShaderWaveform (0 points: no waveform)              */
    int32_t volume_window;                   /* int(0.1*samplerate), audio/module.py:457            */
    int32_t use_mfma;
    int32_t volume_integrate, std_integrate; /* ShaderDynamics.integrate (audio/module.py:413-421)  */
    int32_t length_samples;                  /* ShaderSpectrogram.length_samples = width of its texture (spectrogram.py:272-274);
                                              * <= 1: one column (length = 0). > 1: the scrolling texture — column (k+1) % width is
                                              * rewritten by frame k (:303, 308-311) and every tape frame samples the texture as it
                                              * was when that frame was drawn */
    double precision;                        /* DynamicNumber.precision, dynamics.py:147            */
} sfx_tape_desc;

int sfx_tape_create(sfx_handle plan, sfx_handle audio, const sfx_tape_desc* desc, int max_frames, sfx_handle* tape);
/* A tape for scenes without audio modules: only the frame clock (iTime, iTau, iFrame) varies between the frames of a batch;
 * built with sfx_tape_build(tape, n, NULL, clock, NULL, NULL, NULL), rendered with sfx_render_tape (scene.py:456-479 per frame). */
int sfx_clock_tape_create(sfx_handle ctx, int max_frames, sfx_handle* tape);
int sfx_tape_reset(sfx_handle tape);         /* ShaderDynamics.setup → reset (dynamics.py:273-274)  */
/* Computes frames [0, nframes) of the tape from the running dynamics state (continues across calls). The host arrays are borrowed
 * for the call only (copied to pinned memory of the tape). The work is queued on a stream of the tape's own, into the one of its
 * two banks that the last sfx_render_tape did not read, and the call returns without waiting for the device: the next batch's audio
 * kernels run beside the previous batch's render. sfx_render_tape / sfx_tape_read order themselves after the build they read
 * (events); a build orders itself after the renders of the bank it refills. Batches must be built in frame order. */
int sfx_tape_build(sfx_handle tape, int nframes, const int64_t* tell, const sfx_frame_clock* clock,
                   const sfx_dyn_coeff_f32* spectrogram, const sfx_dyn_coeff_f64* volume,
                   const sfx_dyn_coeff_f64* std);
enum { SFX_TAPE_SPECTROGRAM = 0, SFX_TAPE_WAVEFORM = 1, SFX_TAPE_UNIFORMS = 2, SFX_TAPE_TARGETS = 3, SFX_TAPE_LOUDNESS = 4,
       SFX_TAPE_SCROLL = 5 /* the scrolling texture as each frame sees it: [n][bins][length_samples][channels] f32 */ };
/* Copies tape content to the host for inspection: SPECTROGRAM [n][bins][channels] f32, WAVEFORM
 * [n][points][channels] f32, UNIFORMS [n][8] f32 (iTime iTau iAudioVolume iAudioVolumeIntegral iAudioSTD
 * iSpectrogramOffset iFrame pad), TARGETS (unsmoothed) [n][bins][channels] f32, LOUDNESS [n][2] f32 */
int sfx_tape_read(sfx_handle tape, int what, int frame0, int nframes, void* out, size_t nbytes);
int sfx_tape_destroy(sfx_handle tape);

/* Renders tape frames [frame0, frame0+nframes) with `program` into a dense device buffer of nframes RGB8 images
 * (w*h*3 bytes each, rows bottom-up). ssaa is given times 1000 (scene.ssaa may be fractional, scene.py:372-375).
 * (ssaa, subsample) pairs that sfx_fused_supported() accepts use the fused kernel; any other pair renders the
 * batch in two passes (fragment → RGBA8 iScreen scratch → final.glsl), exactly like shader.py:388-405.
 * iSpectrogram/iWaveform samplers and the audio uniforms come from the tape; everything else from the program's
 * uniform block. */
int sfx_render_tape(sfx_handle program, sfx_handle tape, int frame0, int nframes,
                    int width, int height, int ssaa_x1000, int subsample, void* device_out);

/* Device memory helper for callers without their own allocator */
int sfx_device_alloc(sfx_handle ctx, size_t nbytes, void** ptr);
int sfx_device_free(sfx_handle ctx, void* ptr);
/* asynchronous device-to-device copy on the context's stream (a finished frame into a batch buffer) */
int sfx_device_copy(sfx_handle ctx, void* dst, const void* src, size_t nbytes);
int sfx_device_read(sfx_handle ctx, const void* device_ptr, void* host, size_t nbytes);

/* Peer windows — the gather of a sharded export without a collective and without compute units (no reference equivalent,
 * SURVEY.md §8e "or hipMemcpyPeerAsync, each peer on its own xGMI link"; DESIGN.md §6 "device-sdma"). Rank 0 exports its resident
 * frame buffer (a pointer from sfx_device_alloc) as 64 opaque bytes, passes them to the other processes by any means, each maps the
 * buffer and copies its finished frames to where they belong on a copy stream (SDMA engines), concurrently with its next kernels. */
int sfx_peer_export(sfx_handle ctx, void* device_ptr, void* handle64);
int sfx_peer_open(sfx_handle ctx, const void* handle64, void** device_ptr);
int sfx_peer_close(sfx_handle ctx, void* device_ptr);
/* asynchronous; ordered after what the context's stream holds now. `lane` (0..15) tags the source buffer for sfx_peer_fence. The copy is
 * issued by a thread of the context, up to four in flight: on the context's probed HIP copy streams (hipMemcpyAsync) by default, or —
 * SHADERFLOW_PEER=engine, opt-in since round 6 until it has run between two GPUs — on the two SDMA engines HSA recommends for the (owner of
 * the window, this GPU) pair, named, not drawn (hsa_amd_memory_async_copy_on_engine). A copy whose engine never signals is reported as
 * failed after SHADERFLOW_COPY_TIMEOUT seconds and its signal is retired, never re-armed. */
int sfx_peer_copy(sfx_handle ctx, void* remote_dst, const void* local_src, size_t nbytes, int lane);
int sfx_peer_fence(sfx_handle ctx, int lane);    /* host wait: the lane's last copy has left its source (a pipelined sender asks a step later) */
int sfx_peer_flush(sfx_handle ctx);              /* host wait: every copy issued so far has landed */
/* how the context's peer copies travel: *via_engines 1 = named SDMA engines (engine_ids[2]: their indices), 0 = HIP copy streams, -1 = none
 * issued yet; copies / bytes queued so far. Any pointer may be NULL. */
int sfx_peer_route(sfx_handle ctx, int* via_engines, int* engine_ids, unsigned long long* copies, unsigned long long* bytes);

#ifdef __cplusplus
}
#endif
#endif
