// capi_audio.hip — the C-ABI's audio half (include/shaderflow_hip.h: sfx_flac_*, sfx_audio_*, sfx_stft_*, sfx_filterbank_apply,
// sfx_spectrogram_targets, sfx_waveform_rows, sfx_volume_std, sfx_dynamics_scan*, sfx_tape_* except the render): PCM on the device, the
// STFT plan, and the tape — a batch of frames' audio state built on a stream of its own. The kernels are audio_kernels.hpp's.
// (capi.hip renders FROM a tape through TapeView, host_state.hpp.)

#include "host_state.hpp"
#include "audio_kernels.hpp"
#include "visualizer_kernels.hpp"

#include <algorithm>
#include <vector>

using namespace sf;

#include "flac.inc"

// ---------------------------------------------------------------------------------------------------------
// Audio

struct Audio : Object {
    Context* ctx;
    float* pcm = nullptr;            // planar [channels][samples]
    long samples; int channels, samplerate;
};

extern "C" int sfx_audio_upload(sfx_handle h, const float* interleaved, int64_t samples, int channels, int samplerate, sfx_handle* out) {
    CTX_OR_FAIL(c, h);
    if (!out || samples < 0 || channels < 1 || channels > 8 || (samples > 0 && !interleaved)) return fail(SFX_E_INVALID, "audio of %lld samples x %d channels", (long long)samples, channels);
    USE_DEVICE(c);
    Audio* a = new Audio();
    a->magic = MAGIC_AUDIO; a->ctx = c; a->samples = samples; a->channels = channels; a->samplerate = samplerate;
    std::vector<float> planar((size_t)samples*channels);
    for (int64_t i = 0; i < samples; i++) for (int ch = 0; ch < channels; ch++) planar[(size_t)ch*samples + i] = interleaved[i*channels + ch];
    HIP_TRY(hipMalloc(&a->pcm, planar.size()*sizeof(float) + 16));
    if (!planar.empty()) HIP_TRY(hipMemcpy(a->pcm, planar.data(), planar.size()*sizeof(float), hipMemcpyHostToDevice));
    *out = handle_of(a);
    return SFX_OK;
}

extern "C" int sfx_audio_destroy(sfx_handle h) {
    Audio* a = get<Audio>(h, MAGIC_AUDIO);
    if (!a) return fail(SFX_E_INVALID, "invalid audio handle");
    hipSetDevice(a->ctx->device);
    hipStreamSynchronize(a->ctx->stream);
    hipFree(a->pcm);
    a->magic = 0;
    delete a;
    return SFX_OK;
}

// k-split partial sums of the MFMA filterbank: one per USER of a plan — the plan's own for the per-frame entry points on the context's
// stream, one per tape for its builds on the tape's audio stream — so that a build never shares scratch with a launch on another
// stream (the plan-level buffer was written by both)
struct FilterbankScratch { float* d_partial = nullptr; size_t floats = 0; };

struct Plan : Object {
    Context* ctx;
    int fft_n, window, bins, channels, fft_bins, nnz;
    int fft_size = 0;                // inputs of the transform: 2**fft_n, or int(2**fft_n * sample_rateio) (spectrogram.py:144-146)
    ResampleTap* d_taps = nullptr;   // sample_rateio != 1: where libsamplerate's linear converter reads input sample n of the transform
    int amplitude = 0;               // FourierMagnitude: 0 Power, 1 Amplitude (spectrogram.py:20-26)
    double* d_window = nullptr; double2* d_twiddle = nullptr;
    int *d_indptr = nullptr, *d_indices = nullptr; float* d_data = nullptr;
    float* d_dense = nullptr; int2* d_band = nullptr; int k_pad = 0, row_tiles = 0;
    // scratch that grows on demand
    long* d_tell = nullptr; float* d_power = nullptr; float* d_out = nullptr; int cap_frames = 0;
    FilterbankScratch scratch;                                    // … of the per-frame entry points (the context's stream)
};

static int plan_reserve(Plan* p, int frames) {
    if (frames <= p->cap_frames) return SFX_OK;
    hipStreamSynchronize(p->ctx->stream);
    hipFree(p->d_tell); hipFree(p->d_power); hipFree(p->d_out);
    p->d_tell = nullptr; p->d_power = nullptr; p->d_out = nullptr; p->cap_frames = 0;
    HIP_TRY(hipMalloc(&p->d_tell, sizeof(long)*frames));
    HIP_TRY(hipMalloc(&p->d_power, sizeof(float)*(size_t)frames*p->channels*p->fft_bins));
    HIP_TRY(hipMalloc(&p->d_out, sizeof(float)*(size_t)frames*p->channels*p->bins));
    p->cap_frames = frames;
    return SFX_OK;
}

// A window of the caller's own (spectrogram.py:155-171 multiplies by whatever `self.window(N)` returns, in float64): replaces the
// plan's table; `n` must be the plan's transform size (2**fft_n, or int(2**fft_n * sample_rateio) of a resampled plan).
extern "C" int sfx_stft_plan_window(sfx_handle h, const double* window, int n) {
    Plan* p = get<Plan>(h, MAGIC_PLAN);
    if (!p || !window || n != p->fft_size) return fail(SFX_E_INVALID, "stft plan window: %d values for a plan of %d", n, p ? p->fft_size : 0);
    USE_DEVICE(p->ctx);
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    HIP_TRY(hipMemcpy(p->d_window, window, sizeof(double)*n, hipMemcpyHostToDevice));
    return SFX_OK;
}

static int make_stft_plan(sfx_handle h, int fft_n, int fft_size, const int32_t* tap_a, const int32_t* tap_b, const double* tap_w, int window, int bins, int channels,
                          const int32_t* indptr, const int32_t* indices, const float* data, sfx_handle* out);
extern "C" int sfx_stft_plan(sfx_handle h, int fft_n, int window, int bins, int channels,
                             const int32_t* indptr, const int32_t* indices, const float* data, sfx_handle* out) {
    return make_stft_plan(h, fft_n, (fft_n >= 0 && fft_n < 30) ? (1 << fft_n) : 0, nullptr, nullptr, nullptr, window, bins, channels, indptr, indices, data, out);
}
// `sample_rateio != 1` (spectrogram.py:144-167): the transform takes `fft_size` = int(2**fft_n * ratio) samples, sample n of which is
// (float)(in[tap_a[n]] + tap_w[n]*(in[tap_b[n]] - in[tap_a[n]])) over the last 2**fft_n samples of the ring — the read positions of
// libsamplerate's "linear" converter (samplerate.resample(x, ratio, 'linear'), spectrogram.py:167), which the host derives once per plan
// by running the converter's own float64 position loop (shaderflow_amd/audio/spectrogram.py linear_resample_taps). The built-in windows
// are evaluated for `fft_size`; the filterbank's columns are its fft_size/2 + 1 bins. Power-of-two sizes keep the radix-2 kernel, any
// other size (<= 16 384) takes the float64 DFT sum.
extern "C" int sfx_stft_plan_resampled(sfx_handle h, int fft_n, int fft_size, const int32_t* tap_a, const int32_t* tap_b, const double* tap_w,
                                       int window, int bins, int channels, const int32_t* indptr, const int32_t* indices, const float* data, sfx_handle* out) {
    if (!tap_a || !tap_b || !tap_w) return fail(SFX_E_INVALID, "resampled stft plan: null tap tables");
    return make_stft_plan(h, fft_n, fft_size, tap_a, tap_b, tap_w, window, bins, channels, indptr, indices, data, out);
}
static int make_stft_plan(sfx_handle h, int fft_n, int fft_size, const int32_t* tap_a, const int32_t* tap_b, const double* tap_w, int window, int bins, int channels,
                          const int32_t* indptr, const int32_t* indices, const float* data, sfx_handle* out) {
    CTX_OR_FAIL(c, h);
    if (!out || fft_n < 4 || fft_n > 14 || bins < 1 || channels < 1 || !indptr) return fail(SFX_E_INVALID, "stft plan fft_n=%d bins=%d channels=%d", fft_n, bins, channels);
    if (window < 0 || window > SFX_WINDOW_NONE) return fail(SFX_E_INVALID, "window %d", window);
    if (fft_size < 16 || fft_size > 16384 || (fft_size & 1)) return fail(SFX_E_UNSUPPORTED, "stft transform of %d samples: even sizes from 16 to 16384", fft_size);
    USE_DEVICE(c);
    const int in_size = 1 << fft_n;
    const int N = fft_size, fft_bins = N/2 + 1, nnz = indptr[bins];
    const bool radix2 = (N & (N - 1)) == 0;
    for (int r = 0; r < bins; r++) if (indptr[r] > indptr[r + 1]) return fail(SFX_E_INVALID, "indptr not monotone");
    for (int j = 0; j < nnz; j++) if (indices[j] < 0 || indices[j] >= fft_bins) return fail(SFX_E_INVALID, "column %d outside %d fft bins", indices[j], fft_bins);
    if (tap_a) for (int n = 0; n < N; n++) if (tap_a[n] < 0 || tap_a[n] >= in_size || tap_b[n] < 0 || tap_b[n] >= in_size) return fail(SFX_E_INVALID, "resample tap %d reads outside the %d ring samples", n, in_size);
    Plan* p = new Plan();
    p->magic = MAGIC_PLAN; p->ctx = c; p->fft_n = fft_n; p->fft_size = N; p->window = window; p->bins = bins; p->channels = channels;
    p->fft_bins = fft_bins; p->nnz = nnz;
    // windows: spectrogram.py:92-108 (np.hanning is the symmetric Hann)
    std::vector<double> win(N);
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < N; i++) {
        if (window == SFX_WINDOW_HANNING) win[i] = (N == 1) ? 1.0 : 0.5 + 0.5*::cos(pi*(double)(2*i + 1 - N)/(double)(N - 1));
        else if (window == SFX_WINDOW_HANN_POISSON) win[i] = 0.5*(1.0 - ::cos(2.0*pi*(double)i/(double)N))*::exp(-2.0*::fabs((double)(N - 2*i))/(double)N);
        else win[i] = 1.0;
    }
    // radix-2: exp(-2 pi i k/N) for k < N/2; the DFT sum walks the whole circle
    const int ntw = radix2 ? N/2 : N;
    std::vector<double2> tw(ntw);
    for (int k = 0; k < ntw; k++) { const double ang = -2.0*pi*(double)k/(double)N; tw[k] = make_double2(::cos(ang), ::sin(ang)); }
    HIP_TRY(hipMalloc(&p->d_window, sizeof(double)*N));
    HIP_TRY(hipMalloc(&p->d_twiddle, sizeof(double2)*ntw));
    HIP_TRY(hipMemcpy(p->d_window, win.data(), sizeof(double)*N, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->d_twiddle, tw.data(), sizeof(double2)*ntw, hipMemcpyHostToDevice));
    if (tap_a) {
        std::vector<ResampleTap> taps(N);
        for (int n = 0; n < N; n++) taps[n] = ResampleTap{tap_a[n], tap_b[n], tap_w[n]};
        HIP_TRY(hipMalloc(&p->d_taps, sizeof(ResampleTap)*N));
        HIP_TRY(hipMemcpy(p->d_taps, taps.data(), sizeof(ResampleTap)*N, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&p->d_indptr, sizeof(int)*(bins + 1)));
    HIP_TRY(hipMalloc(&p->d_indices, sizeof(int)*(nnz + 1)));
    HIP_TRY(hipMalloc(&p->d_data, sizeof(float)*(nnz + 1)));
    HIP_TRY(hipMemcpy(p->d_indptr, indptr, sizeof(int)*(bins + 1), hipMemcpyHostToDevice));
    if (nnz) {
        HIP_TRY(hipMemcpy(p->d_indices, indices, sizeof(int)*nnz, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_data, data, sizeof(float)*nnz, hipMemcpyHostToDevice));
    }
    // dense banded copy for the MFMA path: rows padded to 32, k padded to 32, per-row-tile k range
    p->row_tiles = (bins + 31)/32;
    p->k_pad = ((fft_bins + 31)/32)*32;
    std::vector<float> dense((size_t)p->row_tiles*32*p->k_pad, 0.0f);
    std::vector<int2> band(p->row_tiles);
    for (int t = 0; t < p->row_tiles; t++) {
        int lo = p->k_pad, hi = 0;
        for (int r = t*32; r < bins && r < t*32 + 32; r++)
            for (int j = indptr[r]; j < indptr[r + 1]; j++) {
                dense[(size_t)r*p->k_pad + indices[j]] = data[j];
                lo = indices[j] < lo ? indices[j] : lo; hi = indices[j] + 1 > hi ? indices[j] + 1 : hi;
            }
        if (hi <= lo) { lo = 0; hi = 0; }
        band[t] = make_int2((lo/32)*32, ((hi + 31)/32)*32);
    }
    HIP_TRY(hipMalloc(&p->d_dense, sizeof(float)*dense.size()));
    HIP_TRY(hipMalloc(&p->d_band, sizeof(int2)*band.size()));
    HIP_TRY(hipMemcpy(p->d_dense, dense.data(), sizeof(float)*dense.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->d_band, band.data(), sizeof(int2)*band.size(), hipMemcpyHostToDevice));
    if (radix2 && (size_t)(N/2)*sizeof(double2) > 64*1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_stft_power, hipFuncAttributeMaxDynamicSharedMemorySize, (N/2)*(int)sizeof(double2)));
    if (!radix2 && (size_t)N*sizeof(double) > 64*1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_dft_power, hipFuncAttributeMaxDynamicSharedMemorySize, N*(int)sizeof(double)));
    *out = handle_of(p);
    return SFX_OK;
}

extern "C" int sfx_stft_plan_magnitude(sfx_handle h, int magnitude) {
    Plan* p = get<Plan>(h, MAGIC_PLAN);
    if (!p) return fail(SFX_E_INVALID, "invalid plan handle");
    if (magnitude != SFX_MAGNITUDE_POWER && magnitude != SFX_MAGNITUDE_AMPLITUDE) return fail(SFX_E_INVALID, "magnitude %d", magnitude);
    p->amplitude = (magnitude == SFX_MAGNITUDE_AMPLITUDE);
    return SFX_OK;
}

extern "C" int sfx_stft_plan_destroy(sfx_handle h) {
    Plan* p = get<Plan>(h, MAGIC_PLAN);
    if (!p) return fail(SFX_E_INVALID, "invalid plan handle");
    hipSetDevice(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
    hipFree(p->d_taps); hipFree(p->d_window); hipFree(p->d_twiddle); hipFree(p->d_indptr); hipFree(p->d_indices); hipFree(p->d_data);
    hipFree(p->d_dense); hipFree(p->d_band); hipFree(p->d_tell); hipFree(p->d_power); hipFree(p->d_out); hipFree(p->scratch.d_partial);
    p->magic = 0;
    delete p;
    return SFX_OK;
}

static int check_audio(const Plan* p, const Audio* a) {
    if (!p || !a) return fail(SFX_E_INVALID, "invalid plan or audio handle");
    if (p->ctx != a->ctx) return fail(SFX_E_INVALID, "plan and audio belong to different contexts");
    if (a->channels != p->channels) return fail(SFX_E_INVALID, "plan built for %d channels, audio has %d (spectrogram.py:306 hard-codes the reshape)", p->channels, a->channels);
    return SFX_OK;
}

// K3: one wave for up to 256 values plus one for the float64 systems (no barriers), 1024 threads for up to 2048 — and, since
// `spectrogram_bins` is anything the user says (spectrogram.py:184; 1 025 stereo bins already exceed 2 048 values), 4 / 8 / 16 values
// per thread for up to 16 384: the early-out's maximum is still ONE block-wide reduction per frame (the recurrence couples the values
// through it, so the scan stays in one block; at 16 waves a thread may hold 128 registers)
constexpr int DYNAMICS_SCAN_LIMIT = 16384;
template <class... Args> static void launch_dynamics_scan(hipStream_t s, int nframes, int n, Args... args) {
    if (n <= 256) hipLaunchKernelGGL((k_dynamics_scan<64, 4, true>), dim3(1), dim3(128), 0, s, nframes, n, args...);
    else if (n <= 2048) hipLaunchKernelGGL((k_dynamics_scan<1024, 2>), dim3(1), dim3(1024), 0, s, nframes, n, args...);
    else if (n <= 4096) hipLaunchKernelGGL((k_dynamics_scan<1024, 4>), dim3(1), dim3(1024), 0, s, nframes, n, args...);
    else if (n <= 8192) hipLaunchKernelGGL((k_dynamics_scan<1024, 8>), dim3(1), dim3(1024), 0, s, nframes, n, args...);
    else hipLaunchKernelGGL((k_dynamics_scan<1024, 16>), dim3(1), dim3(1024), 0, s, nframes, n, args...);
}

// device-side launches shared by the per-frame entry points and the tape
// `what`: 0 power, 1 amplitude (float32 into d_power), 2 the complex spectrum (float64 pairs into d_power, which then is a double2 buffer)
static void launch_stft(const Plan* p, const Audio* a, const long* d_tell, int frames, float* d_power, hipStream_t s, int what = -1) {
    const int N = p->fft_size, in_size = 1 << p->fft_n;
    if (what < 0) what = p->amplitude;
    if ((N & (N - 1)) == 0)
        hipLaunchKernelGGL(k_stft_power, dim3(frames, p->channels), dim3(256), (N/2)*sizeof(double2), s,
                           a->pcm, a->samples, d_tell, __builtin_ctz((unsigned)N), in_size, p->d_taps, p->d_window, p->d_twiddle, d_power, what);
    else
        hipLaunchKernelGGL(k_dft_power, dim3(frames, p->channels), dim3(256), (size_t)N*sizeof(double), s,
                           a->pcm, a->samples, d_tell, N, in_size, p->d_taps, p->d_window, p->d_twiddle, d_power, what);
}
static void launch_filterbank(Plan* p, FilterbankScratch& scratch, int frames, int use_mfma, const float* d_power, float* d_out, hipStream_t s) {
    const int ncols = frames*p->channels;
    const size_t partial = (size_t)FILTERBANK_SPLITS*p->row_tiles*32*ncols;
    if (use_mfma && scratch.floats < partial) {
        hipStreamSynchronize(s);                                      // the scratch's only user is this stream
        hipFree(scratch.d_partial); scratch.d_partial = nullptr; scratch.floats = 0;
        if (hipMalloc(&scratch.d_partial, partial*sizeof(float)) == hipSuccess) scratch.floats = partial;
        else { (void)hipGetLastError(); use_mfma = 0; }               // out of memory for the scratch: the CSR kernel needs none
    }
    if (use_mfma) {
        hipLaunchKernelGGL(k_filterbank_mfma, dim3((ncols + 31)/32, p->row_tiles, FILTERBANK_SPLITS), dim3(64), 0, s,
                           p->d_dense, p->k_pad, p->d_band, p->fft_bins, ncols, d_power, scratch.d_partial);
        hipLaunchKernelGGL(k_filterbank_reduce, dim3((ncols + 31)/32, (p->bins + 7)/8), dim3(256), 0, s, scratch.d_partial, p->row_tiles*32, p->bins, p->channels, ncols, d_out);
    } else {
        const long total = (long)ncols*p->bins;
        hipLaunchKernelGGL(k_filterbank_csr, dim3((unsigned)((total + 255)/256)), dim3(256), 0, s,
                           p->d_indptr, p->d_indices, p->d_data, p->bins, p->channels, p->fft_bins, ncols, d_power, d_out);
    }
}

extern "C" int sfx_stft_power(sfx_handle hp, sfx_handle ha, const int64_t* tell, int nframes, float* power) {
    Plan* p = get<Plan>(hp, MAGIC_PLAN); Audio* a = get<Audio>(ha, MAGIC_AUDIO);
    int rc = check_audio(p, a);
    if (rc) return rc;
    if (!tell || !power || nframes < 1) return fail(SFX_E_INVALID, "null tell/power or no frames");
    USE_DEVICE(p->ctx);
    if ((rc = plan_reserve(p, nframes))) return rc;
    hipStream_t s = p->ctx->stream;
    HIP_TRY(hipMemcpyAsync(p->d_tell, tell, sizeof(long)*nframes, hipMemcpyHostToDevice, s));
    launch_stft(p, a, p->d_tell, nframes, p->d_power, s);
    if ((rc = launch_status())) return rc;
    HIP_TRY(hipMemcpyAsync(power, p->d_power, sizeof(float)*(size_t)nframes*p->channels*p->fft_bins, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SFX_OK;
}

// `magnitude` callables of the user's own (spectrogram.py:20-41, 169-171 accepts ANY callable on the complex spectrum): the device computes
// np.fft.rfft(window*frame) and hands the float64 pairs over, the host applies the callable, sfx_filterbank_apply takes its float32 result
// through the filterbank. A slow path (two host round trips per call) for an option nothing in the reference's tree uses — but it works.
extern "C" int sfx_stft_spectrum(sfx_handle hp, sfx_handle ha, const int64_t* tell, int nframes, double* spectrum) {
    Plan* p = get<Plan>(hp, MAGIC_PLAN); Audio* a = get<Audio>(ha, MAGIC_AUDIO);
    int rc = check_audio(p, a);
    if (rc) return rc;
    if (!tell || !spectrum || nframes < 1) return fail(SFX_E_INVALID, "null tell/spectrum or no frames");
    USE_DEVICE(p->ctx);
    if ((rc = plan_reserve(p, nframes))) return rc;
    hipStream_t s = p->ctx->stream;
    const size_t bytes = sizeof(double)*2*(size_t)nframes*p->channels*p->fft_bins;
    void* d_spectrum = nullptr;
    HIP_TRY(hipMalloc(&d_spectrum, bytes));
    hipError_t e = hipMemcpyAsync(p->d_tell, tell, sizeof(long)*nframes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) { launch_stft(p, a, p->d_tell, nframes, (float*)d_spectrum, s, 2); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(spectrum, d_spectrum, bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    hipFree(d_spectrum);
    return e == hipSuccess ? SFX_OK : fail(SFX_E_HIP, "stft spectrum: %s", hipGetErrorString(e));
}

extern "C" int sfx_filterbank_apply(sfx_handle hp, const float* magnitudes, int nframes, int use_mfma, float* out) {
    Plan* p = get<Plan>(hp, MAGIC_PLAN);
    if (!p) return fail(SFX_E_INVALID, "invalid plan handle");
    if (!magnitudes || !out || nframes < 1) return fail(SFX_E_INVALID, "null magnitudes/out or no frames");
    USE_DEVICE(p->ctx);
    int rc = plan_reserve(p, nframes);
    if (rc) return rc;
    hipStream_t s = p->ctx->stream;
    HIP_TRY(hipMemcpyAsync(p->d_power, magnitudes, sizeof(float)*(size_t)nframes*p->channels*p->fft_bins, hipMemcpyHostToDevice, s));
    launch_filterbank(p, p->scratch, nframes, use_mfma, p->d_power, p->d_out, s);
    if ((rc = launch_status())) return rc;
    HIP_TRY(hipMemcpyAsync(out, p->d_out, sizeof(float)*(size_t)nframes*p->channels*p->bins, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SFX_OK;
}

extern "C" int sfx_spectrogram_targets(sfx_handle hp, sfx_handle ha, const int64_t* tell, int nframes, int use_mfma, float* out) {
    Plan* p = get<Plan>(hp, MAGIC_PLAN); Audio* a = get<Audio>(ha, MAGIC_AUDIO);
    int rc = check_audio(p, a);
    if (rc) return rc;
    if (!tell || !out || nframes < 1) return fail(SFX_E_INVALID, "null tell/out or no frames");
    USE_DEVICE(p->ctx);
    if ((rc = plan_reserve(p, nframes))) return rc;
    hipStream_t s = p->ctx->stream;
    HIP_TRY(hipMemcpyAsync(p->d_tell, tell, sizeof(long)*nframes, hipMemcpyHostToDevice, s));
    launch_stft(p, a, p->d_tell, nframes, p->d_power, s);
    launch_filterbank(p, p->scratch, nframes, use_mfma, p->d_power, p->d_out, s);
    if ((rc = launch_status())) return rc;
    HIP_TRY(hipMemcpyAsync(out, p->d_out, sizeof(float)*(size_t)nframes*p->channels*p->bins, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SFX_OK;
}

extern "C" int sfx_waveform_rows(sfx_handle ha, const int64_t* tell, int nframes, int chunk, int points, int reducer, float* out) {
    Audio* a = get<Audio>(ha, MAGIC_AUDIO);
    if (!a || !tell || !out || nframes < 1 || chunk < 1 || points < 1) return fail(SFX_E_INVALID, "invalid audio handle or arguments");
    USE_DEVICE(a->ctx);
    hipStream_t s = a->ctx->stream;
    long* d_tell; float* d_rows;
    const size_t n = (size_t)nframes*points*a->channels;
    HIP_TRY(hipMalloc(&d_tell, sizeof(long)*nframes));
    HIP_TRY(hipMalloc(&d_rows, sizeof(float)*n));
    HIP_TRY(hipMemcpyAsync(d_tell, tell, sizeof(long)*nframes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_waveform_rows, dim3((points*a->channels + 3)/4, nframes), dim3(256), 0, s,
                       a->pcm, a->samples, a->channels, d_tell, chunk, points, reducer, d_rows);
    int rc = launch_status();
    if (!rc) { hipMemcpyAsync(out, d_rows, sizeof(float)*n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
    hipFree(d_tell); hipFree(d_rows);
    return rc;
}

extern "C" int sfx_volume_std(sfx_handle ha, const int64_t* tell, int nframes, int window_samples, float* out) {
    Audio* a = get<Audio>(ha, MAGIC_AUDIO);
    if (!a || !tell || !out || nframes < 1 || window_samples < 1) return fail(SFX_E_INVALID, "invalid audio handle or arguments");
    USE_DEVICE(a->ctx);
    hipStream_t s = a->ctx->stream;
    long* d_tell; float* d_out;
    HIP_TRY(hipMalloc(&d_tell, sizeof(long)*nframes));
    HIP_TRY(hipMalloc(&d_out, sizeof(float)*2*nframes));
    HIP_TRY(hipMemcpyAsync(d_tell, tell, sizeof(long)*nframes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_volume_std, dim3(nframes), dim3(256), 0, s, a->pcm, a->samples, a->channels, d_tell, window_samples, d_out);
    int rc = launch_status();
    if (!rc) { hipMemcpyAsync(out, d_out, sizeof(float)*2*nframes, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
    hipFree(d_tell); hipFree(d_out);
    return rc;
}

// DynamicNumber.next over a run of frames, on its own (SURVEY.md §8b last row; dynamics.py:197-250). Host arrays in and out.
extern "C" int sfx_dynamics_scan(sfx_handle h, int nframes, int n, const float* targets, const sfx_dyn_coeff_f32* coeff,
                                 float precision, float* state, float* values) {
    CTX_OR_FAIL(c, h);
    if (nframes < 1 || n < 1 || !targets || !coeff || !state || !values) return fail(SFX_E_INVALID, "dynamics scan: null array or nothing to do");
    if (n > DYNAMICS_SCAN_LIMIT) return fail(SFX_E_UNSUPPORTED, "dynamics scan handles up to %d values per system, got %d", DYNAMICS_SCAN_LIMIT, n);
    USE_DEVICE(c);
    hipStream_t s = c->stream;
    float *d_targets = nullptr, *d_state = nullptr, *d_values = nullptr; DynCoeffF32* d_coeff = nullptr;
    const size_t frame_bytes = sizeof(float)*(size_t)nframes*n;
    const bool ok = hipMalloc(&d_targets, frame_bytes) == hipSuccess && hipMalloc(&d_values, frame_bytes) == hipSuccess &&
                    hipMalloc(&d_state, sizeof(float)*3*n) == hipSuccess && hipMalloc(&d_coeff, sizeof(DynCoeffF32)*nframes) == hipSuccess;
    int rc = ok ? SFX_OK : fail(SFX_E_HIP, "dynamics scan of %d frames x %d values: out of device memory", nframes, n);
    if (!rc) {
        hipMemcpyAsync(d_targets, targets, frame_bytes, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(d_state, state, sizeof(float)*3*n, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(d_coeff, coeff, sizeof(DynCoeffF32)*nframes, hipMemcpyHostToDevice, s);
        launch_dynamics_scan(s, nframes, n, d_targets, d_coeff, precision, d_state, d_values,
                             (const float*)nullptr, (const DynCoeffF64*)nullptr, (const DynCoeffF64*)nullptr, 0.0, 0, 0,
                             (ScalarState*)nullptr, (const FrameClock*)nullptr, (FrameDyn*)nullptr);
        rc = launch_status();
    }
    if (!rc) {
        hipMemcpyAsync(values, d_values, frame_bytes, hipMemcpyDeviceToHost, s);
        hipMemcpyAsync(state, d_state, sizeof(float)*3*n, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) rc = fail(SFX_E_HIP, "dynamics scan: stream synchronisation failed");
    }
    hipFree(d_targets); hipFree(d_values); hipFree(d_state); hipFree(d_coeff);
    return rc;
}

extern "C" int sfx_dynamics_scan_f64(sfx_handle h, int nframes, int nsystems, const double* targets, const sfx_dyn_coeff_f64* coeff,
                                     double precision, int integrate, double* state, double* out) {
    CTX_OR_FAIL(c, h);
    if (nframes < 1 || nsystems < 1 || !targets || !coeff || !state || !out) return fail(SFX_E_INVALID, "dynamics scan: null array or nothing to do");
    static_assert(sizeof(ScalarState) == 4*sizeof(double), "state = value, derivative, previous, integral");
    USE_DEVICE(c);
    hipStream_t s = c->stream;
    double *d_targets = nullptr, *d_out = nullptr; ScalarState* d_state = nullptr; DynCoeffF64* d_coeff = nullptr;
    const size_t count = (size_t)nframes*nsystems;
    const bool ok = hipMalloc(&d_targets, sizeof(double)*count) == hipSuccess && hipMalloc(&d_out, sizeof(double)*3*count) == hipSuccess &&
                    hipMalloc(&d_state, sizeof(ScalarState)*nsystems) == hipSuccess && hipMalloc(&d_coeff, sizeof(DynCoeffF64)*count) == hipSuccess;
    int rc = ok ? SFX_OK : fail(SFX_E_HIP, "dynamics scan of %d frames x %d systems: out of device memory", nframes, nsystems);
    if (!rc) {
        hipMemcpyAsync(d_targets, targets, sizeof(double)*count, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(d_state, state, sizeof(ScalarState)*nsystems, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(d_coeff, coeff, sizeof(DynCoeffF64)*count, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(k_dynamics_scan_f64, dim3((nsystems + 63)/64), dim3(64), 0, s, nframes, nsystems, d_targets, d_coeff, precision, integrate, d_state, d_out);
        rc = launch_status();
    }
    if (!rc) {
        hipMemcpyAsync(out, d_out, sizeof(double)*3*count, hipMemcpyDeviceToHost, s);
        hipMemcpyAsync(state, d_state, sizeof(ScalarState)*nsystems, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) rc = fail(SFX_E_HIP, "dynamics scan: stream synchronisation failed");
    }
    hipFree(d_targets); hipFree(d_out); hipFree(d_state); hipFree(d_coeff);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------
// Tape

// The arrays a batch of frames lives in exist TWICE (two banks): sfx_tape_build fills the bank the last render did not read, on
// the tape's own stream, while the context's stream still renders from the other one — the audio kernels of batch i + 1 (a chain
// of small latency-bound launches, 0.13-0.2 ms) run beside the render of batch i instead of in front of batch i + 1's, and the host
// never waits for a render to hand over its schedule (the borrowed host arrays are copied to pinned memory of the bank).
// Events order the two streams: `built` (recorded after a bank's last audio kernel; renders and reads wait for it), `rendered`
// (recorded after every render from a bank; the build that refills the bank waits for it). The recurrences' state (d_state,
// d_scalars, the scrolling ring) exists once: only the tape's stream touches it, in frame order.
struct TapeBank {
    long* d_tell = nullptr; float *d_power = nullptr, *d_targets = nullptr, *d_columns = nullptr, *d_rows = nullptr, *d_loudness = nullptr;
    FrameDyn* d_dyn = nullptr; DynCoeffF32* d_coeff = nullptr; DynCoeffF64 *d_vol = nullptr, *d_std = nullptr; FrameClock* d_clock = nullptr;
    VisualizerConsts* d_vis = nullptr; float *d_bars = nullptr, *d_scroll = nullptr;
    char* staging = nullptr;         // pinned: the host's schedule arrays of the batch, laid out like d_schedule
    char* d_schedule = nullptr;      // d_tell | d_clock | d_coeff | d_vol | d_std in one allocation: one copy per build
    hipEvent_t built = nullptr, rendered = nullptr;
};
struct Tape : Object {
    Plan* plan; Audio* audio; Context* ctx;
    sfx_tape_desc desc;
    int max_frames, n;               // n = bins*channels
    // the bank the last sfx_tape_build filled (what renders and reads see)
    long* d_tell = nullptr; float *d_power = nullptr, *d_targets = nullptr, *d_columns = nullptr, *d_rows = nullptr, *d_loudness = nullptr;
    FrameDyn* d_dyn = nullptr;
    DynCoeffF32* d_coeff = nullptr; DynCoeffF64 *d_vol = nullptr, *d_std = nullptr; FrameClock* d_clock = nullptr;
    VisualizerConsts* d_vis = nullptr;
    float* d_bars = nullptr;         // sqrt(column/1000) of every frame of the batch (visualizer.frag:45)
    float* d_scroll = nullptr;       // scrolling spectrogram: the texture's state per frame of the batch
    TapeBank bank[2]; int current = 0; bool built_once = false;
    hipStream_t audio_stream = nullptr;
    FilterbankScratch scratch;       // of this tape's builds (audio_stream)
    float* d_state = nullptr; ScalarState* d_scalars = nullptr;
    void* d_screen = nullptr; size_t screen_bytes = 0;   // iScreen scratch of the two-pass path (frames of a batch)
    // scrolling spectrogram (length_samples > 1, spectrogram.py:298-311): ring of the last columns
    int width = 1, ring_frames = 0; long frames_done = 0;
    float* d_ring = nullptr;
};
static void tape_select(Tape* t, int b) {
    const TapeBank& k = t->bank[b];
    t->d_tell = k.d_tell; t->d_power = k.d_power; t->d_targets = k.d_targets; t->d_columns = k.d_columns; t->d_rows = k.d_rows;
    t->d_loudness = k.d_loudness; t->d_dyn = k.d_dyn; t->d_coeff = k.d_coeff; t->d_vol = k.d_vol; t->d_std = k.d_std; t->d_clock = k.d_clock;
    t->d_vis = k.d_vis; t->d_bars = k.d_bars; t->d_scroll = k.d_scroll;
    t->current = b;
}
// offsets of the five schedule arrays in a bank's block (each 16-byte aligned), [5] = the block's size
struct ScheduleLayout { size_t at[6]; };
static ScheduleLayout schedule_layout(int frames) {
    const size_t sizes[5] = {sizeof(long), sizeof(FrameClock), sizeof(DynCoeffF32), sizeof(DynCoeffF64), sizeof(DynCoeffF64)};
    ScheduleLayout l; size_t at = 0;
    for (int i = 0; i < 5; i++) { l.at[i] = at; at += (sizes[i]*(size_t)frames + 15) & ~(size_t)15; }
    l.at[5] = at;
    return l;
}
static size_t tape_staging_bytes(int frames) { return std::max(schedule_layout(frames).at[5], sizeof(FrameDyn)*(size_t)frames); }
// streams, events and pinned staging of both banks; false = out of memory
static bool tape_open_streams(Tape* t) {
    // the audio kernels are small and the render kernel fills the chip: at the highest priority their workgroups take the next free
    // slots instead of queueing behind the render's
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    // (SHADERFLOW_TAPE_PRIORITY=normal: A/B switch for measurements — tools/experiments/timeline_overlap.py)
    const char* priority = getenv("SHADERFLOW_TAPE_PRIORITY");
    if (priority && !strcmp(priority, "normal")) greatest = 0;
    if (hipStreamCreateWithPriority(&t->audio_stream, hipStreamNonBlocking, greatest) != hipSuccess) return false;
    for (TapeBank& k : t->bank) {
        if (hipEventCreateWithFlags(&k.built, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&k.rendered, hipEventDisableTiming) != hipSuccess) return false;
        if (hipHostMalloc((void**)&k.staging, tape_staging_bytes(t->max_frames), hipHostMallocDefault) != hipSuccess) return false;
    }
    return true;
}

extern "C" int sfx_tape_reset(sfx_handle h) {
    Tape* t = get<Tape>(h, MAGIC_TAPE);
    if (!t) return fail(SFX_E_INVALID, "invalid tape handle");
    if (!t->plan) return SFX_OK;                                    // clock tape: no recurrences to reset
    USE_DEVICE(t->ctx);
    HIP_TRY(hipMemsetAsync(t->d_state, 0, sizeof(float)*3*t->n, t->audio_stream));   // in order with the builds before and after it
    HIP_TRY(hipMemsetAsync(t->d_scalars, 0, sizeof(ScalarState)*2, t->audio_stream));
    t->frames_done = 0;                                             // the scrolling texture starts empty again
    return SFX_OK;
}

// A tape without audio (plan == 0 and audio == 0, `ctx_for_clock_tape` says where it lives): only the frame clock varies
// between the frames of a batch — scenes without audio modules (Basic, ShaderToy, RayMarch, the fractals).
extern "C" int sfx_clock_tape_create(sfx_handle hc, int max_frames, sfx_handle* out) {
    CTX_OR_FAIL(c, hc);
    if (!out || max_frames < 1) return fail(SFX_E_INVALID, "null output or no frames");
    USE_DEVICE(c);
    Tape* t = new Tape();
    memset(static_cast<void*>(&t->desc), 0, sizeof t->desc);
    t->magic = MAGIC_TAPE; t->plan = nullptr; t->audio = nullptr; t->ctx = c; t->max_frames = max_frames; t->n = 0;
    bool ok = tape_open_streams(t);
    for (TapeBank& k : t->bank)
        ok = ok && hipMalloc(&k.d_dyn, sizeof(FrameDyn)*max_frames) == hipSuccess && hipMalloc(&k.d_vis, sizeof(VisualizerConsts)*max_frames) == hipSuccess;
    tape_select(t, 0);
    if (!ok) {
        sfx_tape_destroy(handle_of(t));
        return fail(SFX_E_HIP, "clock tape of %d frames: out of device memory", max_frames);
    }
    *out = handle_of(t);
    return SFX_OK;
}

extern "C" int sfx_tape_create(sfx_handle hp, sfx_handle ha, const sfx_tape_desc* desc, int max_frames, sfx_handle* out) {
    Plan* p = get<Plan>(hp, MAGIC_PLAN); Audio* a = get<Audio>(ha, MAGIC_AUDIO);
    int rc = check_audio(p, a);
    if (rc) return rc;
    if (!desc || !out || max_frames < 1) return fail(SFX_E_INVALID, "null desc/output or no frames");
    if (p->bins*p->channels > DYNAMICS_SCAN_LIMIT) return fail(SFX_E_UNSUPPORTED, "dynamics scan handles up to %d spectrogram values, got %d", DYNAMICS_SCAN_LIMIT, p->bins*p->channels);
    USE_DEVICE(p->ctx);
    Tape* t = new Tape();
    t->magic = MAGIC_TAPE; t->plan = p; t->audio = a; t->ctx = p->ctx; t->desc = *desc; t->max_frames = max_frames;
    t->n = p->bins*p->channels;
    const size_t F = max_frames;
    const int pts = desc->points > 0 ? desc->points : 1;
    t->width = desc->length_samples > 1 ? desc->length_samples : 1;
    bool allocated = tape_open_streams(t) &&
        hipMalloc(&t->d_state, sizeof(float)*3*t->n) == hipSuccess &&
        hipMalloc(&t->d_scalars, sizeof(ScalarState)*2) == hipSuccess;
    for (TapeBank& k : t->bank)
        allocated = allocated &&
            hipMalloc((void**)&k.d_schedule, schedule_layout(max_frames).at[5]) == hipSuccess &&
            hipMalloc(&k.d_power, sizeof(float)*F*p->channels*p->fft_bins) == hipSuccess &&
            hipMalloc(&k.d_targets, sizeof(float)*F*t->n) == hipSuccess &&
            hipMalloc(&k.d_columns, sizeof(float)*F*t->n) == hipSuccess &&
            hipMalloc(&k.d_rows, sizeof(float)*F*pts*a->channels) == hipSuccess &&
            hipMalloc(&k.d_loudness, sizeof(float)*F*2) == hipSuccess &&
            hipMalloc(&k.d_dyn, sizeof(FrameDyn)*F) == hipSuccess &&
            hipMalloc(&k.d_vis, sizeof(VisualizerConsts)*F) == hipSuccess &&
            hipMalloc(&k.d_bars, sizeof(float)*F*t->n) == hipSuccess &&
            (t->width <= 1 || hipMalloc(&k.d_scroll, sizeof(float)*F*t->n*t->width) == hipSuccess);
    if (allocated && t->width > 1) {
        t->ring_frames = t->width + max_frames;
        allocated = hipMalloc(&t->d_ring, sizeof(float)*(size_t)t->ring_frames*t->n) == hipSuccess;
    }
    if (allocated) {
        const ScheduleLayout l = schedule_layout(max_frames);
        for (TapeBank& k : t->bank) {
            k.d_tell = (long*)(k.d_schedule + l.at[0]); k.d_clock = (FrameClock*)(k.d_schedule + l.at[1]); k.d_coeff = (DynCoeffF32*)(k.d_schedule + l.at[2]);
            k.d_vol = (DynCoeffF64*)(k.d_schedule + l.at[3]); k.d_std = (DynCoeffF64*)(k.d_schedule + l.at[4]);
        }
    }
    tape_select(t, 0);
    if (!allocated) {
        sfx_tape_destroy(handle_of(t));                             // frees what was allocated (hipFree(nullptr) is a no-op)
        return fail(SFX_E_HIP, "tape of %d frames: out of device memory", max_frames);
    }
    *out = handle_of(t);
    return sfx_tape_reset(*out);
}

extern "C" int sfx_tape_build(sfx_handle h, int nframes, const int64_t* tell, const sfx_frame_clock* clock,
                              const sfx_dyn_coeff_f32* spectrogram, const sfx_dyn_coeff_f64* volume, const sfx_dyn_coeff_f64* std_) {
    Tape* t = get<Tape>(h, MAGIC_TAPE);
    if (!t) return fail(SFX_E_INVALID, "invalid tape handle");
    if (nframes < 1 || nframes > t->max_frames || !clock) return fail(SFX_E_INVALID, "tape build of %d frames (capacity %d) or null clock", nframes, t->max_frames);
    USE_DEVICE(t->ctx);
    // the bank the last render did not read; its previous copy out of the pinned staging is long done (two builds ago) — the wait
    // is there for callers that build without rendering
    const int b = t->built_once ? (t->current ^ 1) : 0;
    TapeBank& k = t->bank[b];
    HIP_TRY(hipEventSynchronize(k.built));
    hipStream_t s = t->audio_stream;
    HIP_TRY(hipStreamWaitEvent(s, k.rendered, 0));                  // the renders that read this bank
    if (!t->plan) {                                                 // clock tape: the per-frame uniforms are the clock itself
        FrameDyn* dyn = (FrameDyn*)k.staging;
        for (int f = 0; f < nframes; f++) {
            memset(&dyn[f], 0, sizeof(FrameDyn));
            dyn[f].iTime = clock[f].iTime; dyn[f].iTau = clock[f].iTau; dyn[f].iSpectrogramOffset = clock[f].iSpectrogramOffset; dyn[f].iFrame = clock[f].iFrame;
        }
        HIP_TRY(hipMemcpyAsync(k.d_dyn, dyn, sizeof(FrameDyn)*nframes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(k.built, s));
        tape_select(t, b); t->built_once = true;
        return SFX_OK;
    }
    if (!tell || !spectrogram || !volume || !std_) return fail(SFX_E_INVALID, "tape build with null audio schedule arrays");
    static_assert(sizeof(sfx_dyn_coeff_f32) == sizeof(DynCoeffF32) && sizeof(sfx_dyn_coeff_f64) == sizeof(DynCoeffF64) && sizeof(sfx_frame_clock) == sizeof(FrameClock), "ABI structs");
    static_assert(sizeof(long) == sizeof(int64_t), "tell");
    Plan* p = t->plan; const Audio* a = t->audio;
    // host arrays are borrowed for the call only: into the bank's pinned staging, from there to the device in ONE copy behind the
    // host's back
    const ScheduleLayout l = schedule_layout(t->max_frames);
    memcpy(k.staging + l.at[0], tell, sizeof(long)*nframes);
    memcpy(k.staging + l.at[1], clock, sizeof(FrameClock)*nframes);
    memcpy(k.staging + l.at[2], spectrogram, sizeof(DynCoeffF32)*nframes);
    memcpy(k.staging + l.at[3], volume, sizeof(DynCoeffF64)*nframes);
    memcpy(k.staging + l.at[4], std_, sizeof(DynCoeffF64)*nframes);
    HIP_TRY(hipMemcpyAsync(k.d_schedule, k.staging, l.at[4] + sizeof(DynCoeffF64)*nframes, hipMemcpyHostToDevice, s));
    launch_stft(p, a, k.d_tell, nframes, k.d_power, s);
    launch_filterbank(p, t->scratch, nframes, t->desc.use_mfma, k.d_power, k.d_targets, s);
    if (t->desc.points > 0)
        hipLaunchKernelGGL(k_waveform_rows, dim3((t->desc.points*a->channels + 3)/4, nframes), dim3(256), 0, s,
                           a->pcm, a->samples, a->channels, k.d_tell, t->desc.chunk_size, t->desc.points, t->desc.reducer, k.d_rows);
    hipLaunchKernelGGL(k_volume_std, dim3(nframes), dim3(256), 0, s, a->pcm, a->samples, a->channels, k.d_tell, t->desc.volume_window, k.d_loudness);
    launch_dynamics_scan(s, nframes, t->n, k.d_targets, k.d_coeff, (float)t->desc.precision,
                         t->d_state, k.d_columns, k.d_loudness, k.d_vol, k.d_std, t->desc.precision,
                         t->desc.volume_integrate, t->desc.std_integrate, t->d_scalars, k.d_clock, k.d_dyn);
    if (t->width > 1) {
        const long count = (long)nframes*t->n;
        hipLaunchKernelGGL(k_spectrogram_ring_store, dim3((unsigned)((count + 255)/256)), dim3(256), 0, s, k.d_columns, nframes, t->n, t->frames_done, t->ring_frames, t->d_ring);
        const long texels = count*t->width;
        hipLaunchKernelGGL(k_spectrogram_scroll, dim3((unsigned)((texels + 255)/256)), dim3(256), 0, s, t->d_ring, t->ring_frames, t->frames_done, nframes,
                           p->bins, p->channels, t->width, k.d_scroll);
    }
    t->frames_done += nframes;
    const int rc = launch_status();
    HIP_TRY(hipEventRecord(k.built, s));
    tape_select(t, b); t->built_once = true;
    return rc;
}

extern "C" int sfx_tape_read(sfx_handle h, int what, int frame0, int nframes, void* out, size_t nbytes) {
    Tape* t = get<Tape>(h, MAGIC_TAPE);
    if (!t || !out) return fail(SFX_E_INVALID, "invalid tape handle or output");
    if (frame0 < 0 || nframes < 1 || frame0 + nframes > t->max_frames) return fail(SFX_E_INVALID, "frames [%d, %d) outside the tape", frame0, frame0 + nframes);
    USE_DEVICE(t->ctx);
    const char* src; size_t per;
    const int pts = t->desc.points > 0 ? t->desc.points : 1;
    if (!t->plan && what != SFX_TAPE_UNIFORMS) return fail(SFX_E_INVALID, "a clock tape holds the per-frame uniforms only");
    switch (what) {
        case SFX_TAPE_SPECTROGRAM: src = (const char*)t->d_columns; per = sizeof(float)*t->n; break;
        case SFX_TAPE_WAVEFORM: src = (const char*)t->d_rows; per = sizeof(float)*pts*t->audio->channels; break;
        case SFX_TAPE_UNIFORMS: src = (const char*)t->d_dyn; per = sizeof(FrameDyn); break;
        case SFX_TAPE_TARGETS: src = (const char*)t->d_targets; per = sizeof(float)*t->n; break;
        case SFX_TAPE_LOUDNESS: src = (const char*)t->d_loudness; per = sizeof(float)*2; break;
        case SFX_TAPE_SCROLL:
            if (t->width <= 1) return fail(SFX_E_INVALID, "the tape has no scrolling spectrogram (length_samples <= 1)");
            src = (const char*)t->d_scroll; per = sizeof(float)*t->n*t->width; break;
        default: return fail(SFX_E_INVALID, "tape section %d", what);
    }
    if (nbytes != per*nframes) return fail(SFX_E_INVALID, "tape read of %zu bytes, section needs %zu", nbytes, per*nframes);
    HIP_TRY(hipStreamWaitEvent(t->ctx->stream, t->bank[t->current].built, 0));
    HIP_TRY(hipMemcpyAsync(out, src + per*frame0, nbytes, hipMemcpyDeviceToHost, t->ctx->stream));
    HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    return SFX_OK;
}

extern "C" int sfx_tape_destroy(sfx_handle h) {
    Tape* t = get<Tape>(h, MAGIC_TAPE);
    if (!t) return fail(SFX_E_INVALID, "invalid tape handle");
    hipSetDevice(t->ctx->device);
    if (t->audio_stream) hipStreamSynchronize(t->audio_stream);
    hipStreamSynchronize(t->ctx->stream);
    for (TapeBank& k : t->bank) {
        hipFree(k.d_schedule); hipFree(k.d_power); hipFree(k.d_targets); hipFree(k.d_columns); hipFree(k.d_rows); hipFree(k.d_loudness);
        hipFree(k.d_dyn); hipFree(k.d_vis); hipFree(k.d_bars); hipFree(k.d_scroll);
        if (k.staging) hipHostFree(k.staging);
        if (k.built) hipEventDestroy(k.built);
        if (k.rendered) hipEventDestroy(k.rendered);
    }
    hipFree(t->d_state); hipFree(t->d_scalars); hipFree(t->d_screen); hipFree(t->d_ring); hipFree(t->scratch.d_partial);
    if (t->audio_stream) hipStreamDestroy(t->audio_stream);
    t->magic = 0;
    delete t;
    return SFX_OK;
}

// what sfx_render_tape (capi.hip) reads of a tape: the bank the last build filled
bool tape_view(sfx_handle h, TapeView* view) {
    Tape* t = get<Tape>(h, MAGIC_TAPE);
    if (!t || !view) return false;
    const TapeBank& k = t->bank[t->current];
    view->ctx = t->ctx; view->max_frames = t->max_frames; view->audio = t->plan != nullptr;
    view->dyn = t->d_dyn; view->vis = t->d_vis; view->built = k.built; view->rendered = k.rendered;
    view->width = t->width; view->values = t->n;
    view->bins = t->plan ? t->plan->bins : 0; view->channels = t->plan ? t->plan->channels : 0;
    view->points = t->desc.points; view->pcm_channels = t->audio ? t->audio->channels : 0;
    view->columns = t->d_columns; view->scroll = t->d_scroll; view->rows = t->d_rows; view->bars = t->d_bars;
    return true;
}

// iScreen scratch of the two-pass path (the frames of a batch), owned by the tape and grown on demand
int tape_screen_scratch(sfx_handle h, size_t bytes, hipStream_t stream, void** screen) {
    Tape* t = get<Tape>(h, MAGIC_TAPE);
    if (!t || !screen) return fail(SFX_E_INVALID, "invalid tape handle");
    if (t->screen_bytes < bytes) {
        HIP_TRY(hipStreamSynchronize(stream));
        hipFree(t->d_screen); t->d_screen = nullptr; t->screen_bytes = 0;
        HIP_TRY(hipMalloc(&t->d_screen, bytes));
        t->screen_bytes = bytes;
    }
    *screen = t->d_screen;
    return SFX_OK;
}
