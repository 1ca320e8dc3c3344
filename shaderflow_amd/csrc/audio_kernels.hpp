// audio_kernels.hpp — gfx950 kernels for the audio half of the path:
//   K1 k_stft_power        window·frame + radix-2 real FFT in float64 + |X|² → float32   (spectrogram.py:155-171)
//   K2 k_filterbank_csr    CSR rows in scipy's summation order, float32                   (spectrogram.py:175-176)
//      k_filterbank_mfma   dense banded GEMM on v_mfma_f32_32x32x2_f32 (batch of frames)
//   K3 k_dynamics_scan     DynamicNumber recurrence over the frames of a batch            (dynamics.py:197-250)
//   K4 k_waveform_rows     sqrt(mean|x|) chunks                                           (waveform.py:80-87)
//   K5 k_volume_std        RMS and standard deviation of the last 0.1 s                  (audio/module.py:457-458)
//
// Data layout in HBM: PCM planar float32 [channels][samples] for the whole file (the reference's 30 s ring,
// audio/module.py:103-111, is a window onto it: after `tell` samples the ring's last n+1 samples are
// stream[tell-n-1 .. tell-1]); power [frame][channel][fft_bins]; spectrogram columns [frame][bin][channel]
// (RG texel order, spectrogram.py:306); waveform rows [frame][point][channel] (waveform.py:86).
#pragma once

#include "glsl.hpp"

namespace sf {

__device__ __forceinline__ float stream_at(const float* pcm, long total, int c, long idx) {
    return (idx < 0 || idx >= total) ? 0.0f : pcm[(long)c*total + idx];       // zeros before the file starts
}

// ---- K1 ---------------------------------------------------------------------------------------------------
// One block per (frame, channel). The N real samples are packed as N/2 complex numbers, transformed with an
// in-LDS radix-2 decimation-in-time FFT in float64 (numpy computes window*data and the rfft in float64 and only
// then casts, spectrogram.py:169-171), and split into the N/2+1 real-input bins.
// twiddle[k] = exp(-2*pi*i*k/N), k < N/2, and window[n] are float64 tables built by the host.
//
// `sample_rateio != 1` (spectrogram.py:144-167): the transform's N = int(2**fft_n * ratio) inputs are the last `in_size` = 2**fft_n
// samples of the ring RESAMPLED by libsamplerate's "linear" converter. That converter is a one-sample-delayed linear interpolation
// whose read positions depend on the sizes and the ratio alone (src_linear.c accumulates input_index += 1/ratio in float64), so the
// host runs its position loop once per plan and the device applies the taps: out[n] = (float)(in[a] + w*(in[b] - in[a])) in float64.
struct ResampleTap { int a, b; double w; };
__device__ __forceinline__ double stft_input(const float* pcm, long total, int c, long first, int n, const ResampleTap* __restrict__ taps) {
    if (!taps) return (double)stream_at(pcm, total, c, first + n);
    const ResampleTap t = taps[n];
    const double a = (double)stream_at(pcm, total, c, first + t.a), b = (double)stream_at(pcm, total, c, first + t.b);
    return (double)(float)(a + t.w*(b - a));                          // the converter writes float32 samples
}

__global__ __launch_bounds__(256) void k_stft_power(const float* __restrict__ pcm, long total, const long* __restrict__ tell,
                                                    int fft_n, int in_size, const ResampleTap* __restrict__ taps, const double* __restrict__ window,
                                                    const double2* __restrict__ twiddle, float* __restrict__ power, int amplitude) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* z = (double2*)smem;
    const int N = 1 << fft_n, M = N >> 1, logM = fft_n - 1;
    const int frame = blockIdx.x, c = blockIdx.y, channels = gridDim.y;
    const long first = tell[frame] - in_size - 1;                   // audio/module.py:137-138
    for (int n = threadIdx.x; n < M; n += blockDim.x) {
        const double re = window[2*n]*stft_input(pcm, total, c, first, 2*n, taps);
        const double im = window[2*n + 1]*stft_input(pcm, total, c, first, 2*n + 1, taps);
        z[__brev((unsigned)n) >> (32 - logM)] = make_double2(re, im);
    }
    __syncthreads();
    for (int s = 1; s <= logM; s++) {
        const int half = 1 << (s - 1);
        const int tw_stride = N >> s;                               // W_len^k = W_N^(k*N/len), len = 2^s
        for (int b = threadIdx.x; b < (M >> 1); b += blockDim.x) {
            const int k = b & (half - 1);
            const int i0 = ((b >> (s - 1)) << s) + k, i1 = i0 + half;
            const double2 w = twiddle[k*tw_stride];
            const double2 a = z[i0], v = z[i1];
            const double xr = v.x*w.x - v.y*w.y, xi = v.x*w.y + v.y*w.x;
            z[i0] = make_double2(a.x + xr, a.y + xi);
            z[i1] = make_double2(a.x - xr, a.y - xi);
        }
        __syncthreads();
    }
    float* out = power + ((long)frame*channels + c)*(M + 1);
    for (int k = threadIdx.x; k <= M; k += blockDim.x) {
        const double2 a = z[k & (M - 1)], b = z[(M - k) & (M - 1)];
        const double er = 0.5*(a.x + b.x), ei = 0.5*(a.y - b.y);    // E = (Z[k] + conj(Z[M-k]))/2
        const double orr = 0.5*(a.y + b.y), oi = -0.5*(a.x - b.x);  // O = -i(Z[k] - conj(Z[M-k]))/2
        double2 w = (k < M) ? twiddle[k] : make_double2(-1.0, 0.0);
        const double xr = er + (orr*w.x - oi*w.y), xi = ei + (orr*w.y + oi*w.x);
        // FourierMagnitude.Power = (x*conj(x)).real, spectrogram.py:25-26; .Amplitude = np.abs(x) = hypot in float64, :22-23
        // (amplitude == 2: the COMPLEX spectrum itself, float64 pairs — sfx_stft_spectrum: a `magnitude` callable of the user's own runs on the host)
        if (amplitude == 2) ((double2*)power)[((long)frame*channels + c)*(M + 1) + k] = make_double2(xr, xi);
        else out[k] = amplitude ? (float)::hypot(xr, xi) : (float)(xr*xr + xi*xi);
    }
}

// The same for a transform size that is NOT a power of two (sample_rateio = 1.5: 6 144 inputs): the plain DFT sum in float64, which is
// what np.fft.rfft computes to rounding. The windowed inputs of the (frame, channel) in LDS; a thread owns bins k, k + 256, … and
// walks the unit circle by table: twiddle[j] = exp(-2*pi*i*j/N), j < N, read at j = k*n mod N, kept by adding k and wrapping — no
// angle is ever multiplied out, so the sum's error is that of N additions (~1e-13 relative), far inside the path's 1e-5. O(N^2) per
// block; 600 blocks of 19 M multiply-adds per 300-frame batch at N = 6 144: a few milliseconds, on an option nothing in the
// reference's tree uses.
__global__ __launch_bounds__(256) void k_dft_power(const float* __restrict__ pcm, long total, const long* __restrict__ tell, int N, int in_size,
                                                   const ResampleTap* __restrict__ taps, const double* __restrict__ window,
                                                   const double2* __restrict__ twiddle /* [N] */, float* __restrict__ power, int amplitude) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* x = (double*)smem;
    const int frame = blockIdx.x, c = blockIdx.y, channels = gridDim.y;
    const long first = tell[frame] - in_size - 1;
    for (int n = threadIdx.x; n < N; n += blockDim.x) x[n] = window[n]*stft_input(pcm, total, c, first, n, taps);
    __syncthreads();
    const int bins = N/2 + 1;
    float* out = power + ((long)frame*channels + c)*bins;
    for (int k = threadIdx.x; k < bins; k += blockDim.x) {
        double re = 0.0, im = 0.0;
        int j = 0;
        for (int n = 0; n < N; n++) {
            const double2 w = twiddle[j];
            re = fma(x[n], w.x, re); im = fma(x[n], w.y, im);
            j += k; if (j >= N) j -= N;
        }
        if (amplitude == 2) ((double2*)power)[((long)frame*channels + c)*bins + k] = make_double2(re, im);
        else out[k] = amplitude ? (float)::hypot(re, im) : (float)(re*re + im*im);
    }
}

// ---- K2 (CSR) ---------------------------------------------------------------------------------------------
// y[row,:] += a*x[col,:] entry by entry in index order, separate multiply and add: scipy's csr_matvecs.
__global__ void k_filterbank_csr(const int* __restrict__ indptr, const int* __restrict__ indices, const float* __restrict__ data,
                                 int bins, int channels, int fft_bins, int ncols, const float* __restrict__ power,
                                 float* __restrict__ out) {
    const long t = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (t >= (long)ncols*bins) return;
    const int col = (int)(t / bins), r = (int)(t % bins);            // col = frame*channels + channel
    const float* p = power + (long)col*fft_bins;
    float acc = 0.0f;
    for (int j = indptr[r]; j < indptr[r + 1]; j++) {
        const float prod = data[j]*p[indices[j]];
        acc = acc + prod;
    }
    out[((long)(col / channels)*bins + r)*channels + (col % channels)] = acc;
}

// ---- K2 (MFMA) --------------------------------------------------------------------------------------------
// out[bin][col] = sum_k A[bin][k]*P[col][k] with the dense filterbank A (bins_pad x k_pad, zero padded) and
// only the k range [kbeg, kend) that holds non-zeros for the 32-bin row tile (the matrix is banded: 3-4
// non-zeros per row, spectrogram.py:194-224). Exact float32 products, fma chain in k order.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// One wave per (32 columns, 32-bin row tile, k split): the band of the row tile is cut into FILTERBANK_SPLITS contiguous runs of
// 32-wide chunks, every wave feeds its chunks straight from global memory (L2-resident: 1 MB of power per 60 frames) into the matrix
// pipe — lanes 0-31 supply the even k of a chunk, lanes 32-63 the odd ones, as v_mfma_f32_32x32x2_f32 wants them — and writes its
// 32x32 partial sums; k_filterbank_reduce adds the splits in a fixed order. 4 x 4 x 8 = 128 waves instead of 16 that walked their
// whole band through LDS (209 µs per 60 frames in round 1).
constexpr int FILTERBANK_SPLITS = 8;

__global__ __launch_bounds__(64) void k_filterbank_mfma(const float* __restrict__ A, int k_pad, const int2* __restrict__ band,
                                                        int fft_bins, int ncols, const float* __restrict__ power,
                                                        float* __restrict__ partial /* [split][row_tiles*32][ncols] */) {
    const int tile_row = blockIdx.y, col0 = blockIdx.x*32, split = blockIdx.z;
    const int lane = threadIdx.x & 63, half = lane >> 5, l = lane & 31;
    const int2 kr = band[tile_row];
    const int chunks = (kr.y - kr.x)/32, per = (chunks + FILTERBANK_SPLITS - 1)/FILTERBANK_SPLITS;
    const int first = split*per, last = min(chunks, first + per);
    f32x16 acc = {0};
    const int col = col0 + l;
    const float* a_row = A + (long)(tile_row*32 + l)*k_pad;
    const float* b_row = power + (long)min(col, ncols - 1)*fft_bins;
    for (int c = first; c < last; c++) {
        const int kc = kr.x + c*32 + half;
        float a[16], b[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int k = kc + 2*q;
            a[q] = a_row[k];                                          // zero padded to k_pad
            b[q] = (col < ncols && k < fft_bins) ? b_row[k] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 16; q++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
    }
    if (col < ncols) {
        float* out = partial + ((long)split*gridDim.y*32)*ncols;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int r = tile_row*32 + (reg & 3) + 8*(reg >> 2) + 4*half;
            out[(long)r*ncols + col] = acc[reg];
        }
    }
}

// A block adds the splits of a 32-column x 8-bin tile — read along the columns, as k_filterbank_mfma wrote them, one output per thread
// with its eight partials in flight together — and turns the tile in LDS so that it leaves along the bins, the order of the RG texel
// rows (spectrogram.py:306). (Round 4's version read eight partials a cache line apart per lane, one after the other: 65 us per 300
// frames, more than the CSR kernel it was meant to beat; 1 024 outputs per block still took 48 us: too few loads in flight.)
__global__ __launch_bounds__(256) void k_filterbank_reduce(const float* __restrict__ partial, int rows_pad, int bins, int channels, int ncols, float* __restrict__ out) {
    __shared__ float tile[8][33];
    const int c0 = blockIdx.x*32, r0 = blockIdx.y*8;
    {
        const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
        const int col = c0 + cl, r = r0 + rl;
        float part[FILTERBANK_SPLITS];
#pragma unroll
        for (int split = 0; split < FILTERBANK_SPLITS; split++) part[split] = (col < ncols && r < bins) ? partial[((long)split*rows_pad + r)*ncols + col] : 0.0f;
        float sum = 0.0f;
#pragma unroll
        for (int split = 0; split < FILTERBANK_SPLITS; split++) sum = sum + part[split];      // the splits in their fixed order
        tile[rl][cl] = sum;
    }
    __syncthreads();
    {
        const int rl = threadIdx.x & 7, cl = threadIdx.x >> 3;
        const int col = c0 + cl, r = r0 + rl;
        if (col < ncols && r < bins) out[((long)(col / channels)*bins + r)*channels + (col % channels)] = tile[rl][cl];
    }
}

// ---- K4 ---------------------------------------------------------------------------------------------------
// One wave per (frame, point, channel) chunk of `chunk` samples.
__device__ __forceinline__ float wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__global__ __launch_bounds__(256) void k_waveform_rows(const float* __restrict__ pcm, long total, int channels,
                                                       const long* __restrict__ tell, int chunk, int points, int reducer,
                                                       float* __restrict__ rows) {
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x*4 + (threadIdx.x >> 6);        // (frame, point, channel)
    const long nw = (long)gridDim.y*points*channels;
    (void)nw;
    const int frame = blockIdx.y;
    if (wid >= (long)points*channels) return;
    const int p = (int)(wid / channels), c = (int)(wid % channels);
    const long t = tell[frame];
    const long offset = t % chunk;                                   // waveform.py:71-73
    const long base = t - ((long)chunk*points + offset + 1) + (long)p*chunk;   // waveform.py:81-83
    float result;
    if (reducer == 0) {                                              // sqrt(mean(|x|)), waveform.py:15-16
        float s = 0.0f;
        for (int i = lane; i < chunk; i += 64) s += fabsf(stream_at(pcm, total, c, base + i));
        result = sqrtf(wave_sum(s)/(float)chunk);
    } else if (reducer == 1) {                                       // sqrt(sqrt(mean(x²))*2**0.5), :18-19
        float s = 0.0f;
        for (int i = lane; i < chunk; i += 64) { const float x = stream_at(pcm, total, c, base + i); s += x*x; }
        result = sqrtf(sqrtf(wave_sum(s)/(float)chunk)*(float)1.4142135623730951);
    } else {                                                         // sqrt(std(x)), :21-22
        float s = 0.0f;
        for (int i = lane; i < chunk; i += 64) s += stream_at(pcm, total, c, base + i);
        const float mean = wave_sum(s)/(float)chunk;
        float q = 0.0f;
        for (int i = lane; i < chunk; i += 64) { const float d = stream_at(pcm, total, c, base + i) - mean; q += d*d; }
        result = sqrtf(sqrtf(wave_sum(q)/(float)chunk));
    }
    if (lane == 0) rows[((long)frame*points + p)*channels + c] = result;
}

// ---- K5 ---------------------------------------------------------------------------------------------------
// One block per frame over channels*n samples: volume target = 2*sqrt(mean(x²))*sqrt(2) and std target
// = sqrt(mean((x-mean)²)), rounded to float32 at the points numpy rounds (audio/module.py:74-75,457-458).
__global__ __launch_bounds__(256) void k_volume_std(const float* __restrict__ pcm, long total, int channels,
                                                    const long* __restrict__ tell, int n, float* __restrict__ out) {
    __shared__ double red[2][4];
    __shared__ double mean_sh;
    const int frame = blockIdx.x;
    const long first = tell[frame] - n - 1;
    const int count = n*channels;
    double s1 = 0.0, s2 = 0.0;
    for (int e = threadIdx.x; e < count; e += 256) {
        const double x = (double)stream_at(pcm, total, e / n, first + (e % n));
        s1 += x; s2 += x*x;
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double sum = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const double sq = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const float rms = sqrtf((float)(sq/(double)count));
        out[2*frame] = (2.0f*rms)*(float)1.4142135623730951;
        mean_sh = (double)(float)(sum/(double)count);               // numpy's arrmean is float32
    }
    __syncthreads();
    const double mean = mean_sh;
    double q = 0.0;
    for (int e = threadIdx.x; e < count; e += 256) {
        const float d = stream_at(pcm, total, e / n, first + (e % n)) - (float)mean;   // x - arrmean in float32
        q += (double)(d*d);
    }
    q = wave_sum(q);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) out[2*frame + 1] = sqrtf((float)((red[0][0] + red[0][1] + red[0][2] + red[0][3])/(double)count));
}

// ---- K3 ---------------------------------------------------------------------------------------------------
// The recurrence is sequential in time (an IIR with a data-dependent early-out, dynamics.py:222-225), so one
// block walks the frames of the batch; the per-frame python scalars (dt, k1, k2, k3) come from the host,
// already rounded to float32 the way numpy does for a float32 array (NEP 50).
struct DynCoeffF32 { float dt, k1, k2, k3; };
struct DynCoeffF64 { double dt, k1, k2, k3; };
struct FrameClock { float iTime, iTau, iSpectrogramOffset; int iFrame; };
struct ScalarState { double value, derivative, previous, integral; };

__device__ __forceinline__ void scalar_step(ScalarState& s, double target, const DynCoeffF64& c, double precision, int integrate) {
    if (c.dt == 0.0) return;                                          // dynamics.py:210-211
    if (fabs(target - s.value) < precision) {                         // :222-225
        if (integrate) s.integral += (s.value*c.dt);
        return;
    }
    const double velocity = (target - s.previous)/c.dt;               // :228-229
    s.previous = target;
    s.value += (s.derivative*c.dt);                                   // :245
    const double acceleration = (target + c.k3*velocity - s.value - c.k1*s.derivative)/c.k2;   // :246
    s.derivative += (acceleration*c.dt);                              // :247
    if (integrate) s.integral += (s.value*c.dt);                      // :248-249
}

// THREADS x PER >= n values. One wave (THREADS == 64, up to 256 values: the 115-bin stereo spectrogram of the benchmark) needs no
// barrier at all — the early-out's maximum is a wave reduction; larger states take 1024 threads and two barriers per frame. The
// frame loop is latency bound (a batch is 60 dependent steps of a few hundred values), so the next frame's targets and
// coefficients are requested before the current frame is worked on, and the float64 volume/std systems live in registers for the
// whole batch instead of a global round trip per frame (117 -> see DESIGN §7 µs per 60 frames).
// KEEPER_WAVE: the float64 volume/std systems and the uniforms of a frame are stepped by lane 0 of an extra wave of their own — both
// halves are chains of dependent operations (IEEE divisions in float32 here, in float64 there) that a single wave would run one after
// the other; they share nothing, so no barrier is needed between them.
template <int THREADS, int PER, bool KEEPER_WAVE = false>
__global__ __launch_bounds__(THREADS + (KEEPER_WAVE ? 64 : 0)) void k_dynamics_scan(int nframes, int n /* bins*channels */,
                                                        const float* __restrict__ targets,     // [frame][n]
                                                        const DynCoeffF32* __restrict__ coeff,
                                                        float precision,
                                                        float* __restrict__ state,             // value | derivative | previous, n each
                                                        float* __restrict__ columns,           // [frame][n]
                                                        const float* __restrict__ loudness,    // [frame][2] volume/std targets
                                                        const DynCoeffF64* __restrict__ vol_coeff, const DynCoeffF64* __restrict__ std_coeff,
                                                        double scalar_precision, int vol_integrate, int std_integrate,
                                                        ScalarState* __restrict__ scalars,     // [2]
                                                        const FrameClock* __restrict__ clock, FrameDyn* __restrict__ dyn) {
    constexpr int WAVES = THREADS/64;
    if (KEEPER_WAVE && threadIdx.x >= THREADS) n = 0;                 // the keeper's wave holds no values
    __shared__ float red[WAVES > 1 ? WAVES : 1];
    __shared__ int skip_shared;
    float value[PER], deriv[PER], prev[PER], target[PER], upcoming[PER];
#pragma unroll
    for (int e = 0; e < PER; e++) {
        const int i = threadIdx.x + e*THREADS;
        value[e] = (i < n) ? state[i] : 0.0f;
        deriv[e] = (i < n) ? state[n + i] : 0.0f;
        prev[e] = (i < n) ? state[2*n + i] : 0.0f;
        upcoming[e] = (i < n && nframes > 0) ? targets[i] : 0.0f;
    }
    const bool keeper = threadIdx.x == (KEEPER_WAVE ? THREADS : 0) && dyn;   // the thread that steps the float64 systems and writes the uniforms
    ScalarState v{}, s{};
    if (keeper) { v = scalars[0]; s = scalars[1]; }
    DynCoeffF32 c_next = nframes > 0 ? coeff[0] : DynCoeffF32{};
    // the keeper's per-frame inputs, requested one frame ahead as well
    FrameClock clock_next{}; float loud_next[2] = {0.0f, 0.0f}; DynCoeffF64 vol_next{}, std_next{};
    auto keeper_fetch = [&](int f) {
        clock_next = clock[f];
        if (loudness) { loud_next[0] = loudness[2*f]; loud_next[1] = loudness[2*f + 1]; vol_next = vol_coeff[f]; std_next = std_coeff[f]; }
    };
    if (keeper && nframes > 0) keeper_fetch(0);
    for (int f = 0; f < nframes; f++) {
        const DynCoeffF32 c = c_next;
        const FrameClock clock_now = clock_next; const float loud_now[2] = {loud_next[0], loud_next[1]}; const DynCoeffF64 vol_now = vol_next, std_now = std_next;
        if (keeper && f + 1 < nframes) keeper_fetch(f + 1);
        float worst = 0.0f;
#pragma unroll
        for (int e = 0; e < PER; e++) {
            const int i = threadIdx.x + e*THREADS;
            target[e] = upcoming[e];
            if (i < n) worst = fmaxf(worst, fabsf(target[e] - value[e]));
        }
        if (f + 1 < nframes) {                                        // in flight while this frame is worked on
            c_next = coeff[f + 1];
#pragma unroll
            for (int e = 0; e < PER; e++) {
                const int i = threadIdx.x + e*THREADS;
                upcoming[e] = (i < n) ? targets[(long)(f + 1)*n + i] : 0.0f;
            }
        }
        if (c.dt != 0.0f) {
            for (int m = 32; m >= 1; m >>= 1) worst = fmaxf(worst, __shfl_xor(worst, m));
            bool skip;
            if constexpr (WAVES == 1) {
                skip = worst < precision;                             // np.abs(target - value).max() < precision
            } else {
                if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = worst;
                __syncthreads();
                if (threadIdx.x == 0) {
                    float w = red[0];
                    for (int k = 1; k < WAVES; k++) w = fmaxf(w, red[k]);
                    skip_shared = (w < precision) ? 1 : 0;
                }
                __syncthreads();
                skip = skip_shared != 0;
            }
            if (!skip) {
#pragma unroll
                for (int e = 0; e < PER; e++) {
                    const float velocity = (target[e] - prev[e])/c.dt;
                    prev[e] = target[e];
                    value[e] = value[e] + (deriv[e]*c.dt);
                    const float accel = (((target[e] + (c.k3*velocity)) - value[e]) - (c.k1*deriv[e]))/c.k2;
                    deriv[e] = deriv[e] + (accel*c.dt);
                }
            }
            if constexpr (WAVES > 1) __syncthreads();
        }
#pragma unroll
        for (int e = 0; e < PER; e++) {
            const int i = threadIdx.x + e*THREADS;
            if (i < n) columns[(long)f*n + i] = value[e];
        }
        if (keeper) {
            if (loudness) {
                scalar_step(v, (double)loud_now[0], vol_now, scalar_precision, vol_integrate);
                scalar_step(s, (double)loud_now[1], std_now, scalar_precision, std_integrate);
            }
            FrameDyn d;
            d.iTime = clock_now.iTime; d.iTau = clock_now.iTau; d.iFrame = clock_now.iFrame;
            d.iSpectrogramOffset = clock_now.iSpectrogramOffset;
            d.iAudioVolume = (float)v.value; d.iAudioVolumeIntegral = (float)v.integral; d.iAudioSTD = (float)s.value;
            d.pad = 0;
            dyn[f] = d;
        }
    }
    if (keeper) { scalars[0] = v; scalars[1] = s; }
#pragma unroll
    for (int e = 0; e < PER; e++) {
        const int i = threadIdx.x + e*THREADS;
        if (i < n) { state[i] = value[e]; state[n + i] = deriv[e]; state[2*n + i] = prev[e]; }
    }
}

// ---- scrolling spectrogram texture (spectrogram.py:298-311) -------------------------------------------------
// ShaderSpectrogram keeps a texture of `width` = length*fps columns and overwrites column (offset+1) % width every frame.
// A batch renders its frames concurrently, so every frame gets the texture AS IT WAS when that frame was drawn:
// scroll[f][bin][col][channel] = the column the most recent frame k' <= k with (k'+1) % width == col wrote (zeros before
// any did), k = first_frame + f. `ring` holds the smoothed columns of the last `ring_frames` >= width frames by
// absolute frame index modulo ring_frames.
__global__ void k_spectrogram_ring_store(const float* __restrict__ columns, int nframes, int n, long first_frame, int ring_frames,
                                         float* __restrict__ ring) {
    const long t = (long)blockIdx.x*blockDim.x + threadIdx.x;
    if (t >= (long)nframes*n) return;
    const long f = t / n, e = t % n;
    ring[((first_frame + f) % ring_frames)*n + e] = columns[t];
}
__global__ void k_spectrogram_scroll(const float* __restrict__ ring, int ring_frames, long first_frame, int nframes,
                                     int bins, int channels, int width, float* __restrict__ scroll) {
    const long t = (long)blockIdx.x*blockDim.x + threadIdx.x;
    const long per_frame = (long)bins*width*channels;
    if (t >= per_frame*nframes) return;
    const long f = t / per_frame; long r = t % per_frame;
    const int bin = (int)(r / ((long)width*channels)); r %= (long)width*channels;
    const int col = (int)(r / channels), ch = (int)(r % channels);
    const long k = first_frame + f;
    const long age = (((k + 1 - col) % width) + width) % width;          // frames since column `col` was written
    const long writer = k - age;
    scroll[t] = (writer >= 0) ? ring[((writer % ring_frames)*bins + bin)*channels + ch] : 0.0f;
}

// Scalar float64 systems alone (volume, std, the camera's nine systems — dynamics.py:197-250 with python floats): one thread per
// system walks the frames; out[frame][system] = {value, integral, derivative}.
__global__ void k_dynamics_scan_f64(int nframes, int nsystems, const double* __restrict__ targets /* [frame][system] */,
                                    const DynCoeffF64* __restrict__ coeff /* [system][frame] */, double precision, int integrate,
                                    ScalarState* __restrict__ state, double* __restrict__ out) {
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= nsystems) return;
    ScalarState s = state[k];
    for (int f = 0; f < nframes; f++) {
        scalar_step(s, targets[(long)f*nsystems + k], coeff[(long)k*nframes + f], precision, integrate);
        double* o = out + ((long)f*nsystems + k)*3;
        o[0] = s.value; o[1] = s.integral; o[2] = s.derivative;
    }
    state[k] = s;
}

}  // namespace sf
