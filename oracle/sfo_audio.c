/*
 * ORACLE (test infrastructure, never shipped, never on the product path).
 * Audio half: CPU restatement of the reference's numpy code. PINNED by tests/golden/ fixtures.
 * Citations are file:line in /root/reference (BrokenSource/ShaderFlow v0.11.3).
 */
#include "sfo.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---------------------------------------------------------------------------------------------- */
/* Frame clock: scheduler.py:87-89 (freewheel: started = 0, last_call = -period, next_call = 0),
 * :152-173 (now = next_call; dt = now - last_call; while next_call <= now: next_call += period),
 * and scene.py:475-479 (dt/rdt/time are stored AFTER the modules of the frame ran). */
void sfo_clock(double fps, double speed, int frames, double* time_used, double* dt_used, double* rdt_used) {
    double period = 1.0/fps;
    double last_call = 0.0 - period;
    double next_call = 0.0;
    double time = 0.0, dt = 0.0, rdt = 0.0;
    for (int k = 0; k < frames; k++) {
        time_used[k] = time; dt_used[k] = dt; rdt_used[k] = rdt;
        double now = next_call;
        double task_dt = now - last_call;
        last_call = now;
        /* scene.py:476 sets vsync.fps = fps each frame: period is recomputed, same value */
        while (next_call <= now) next_call += period;
        dt = task_dt*speed;
        rdt = task_dt;
        time += dt;
    }
}

/* ffmpeg.py:1311-1330: target += chunk; length = (target - read/Bps)*Bps; round to blocks; at least
 * one block; a short read ends the stream. Python round() is ties-to-even → rint(). */
void sfo_reader(const double* rdt_used, int frames, int samplerate, int channels,
                int64_t total_samples, int32_t* lengths, int64_t* tell) {
    const int64_t block = 4*(int64_t)channels;                 /* pcm_f32le, ffmpeg.py:1244,1270-1271 */
    const int64_t bps = block*samplerate;
    const int64_t total_bytes = total_samples*block;
    double target = 0.0;
    int64_t read = 0, t = 0;
    int dry = 0;
    for (int k = 0; k < frames; k++) {
        int32_t got = 0;
        if (!dry) {
            target += rdt_used[k];                             /* audio/module.py:450 */
            double length = (target - (double)read/(double)bps)*(double)bps;
            int64_t bytes = block*(int64_t)rint(length/(double)block);
            if (bytes < block) bytes = block;
            if (read + bytes > total_bytes) bytes = total_bytes - read;
            if (bytes <= 0) { dry = 1; }                       /* `if len(data) == 0: break` */
            else { read += bytes; got = (int32_t)(bytes/block); }
        }
        t += got;
        lengths[k] = got;
        tell[k] = t;
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* Windows: spectrogram.py:92-98 (Hann-Poisson), :100-103 (np.hanning = symmetric Hann), :105-108 */
void sfo_window(int kind, int n, double* out) {
    for (int i = 0; i < n; i++) {
        if (kind == SFO_WINDOW_HANNING) {
            /* numpy.hanning(M): 0.5 + 0.5*cos(pi*k/(M-1)), k = 1-M, 3-M, …, M-1 */
            out[i] = (n == 1) ? 1.0 : 0.5 + 0.5*cos(M_PI*(double)(2*i + 1 - n)/(double)(n - 1));
        } else if (kind == SFO_WINDOW_HANN_POISSON) {
            double a = 0.5*(1.0 - cos(2.0*M_PI*(double)i/(double)n));
            double b = exp(-2.0*fabs((double)(n - 2*i))/(double)n);
            out[i] = a*b;
        } else {
            out[i] = 1.0;
        }
    }
}

/* Iterative radix-2 complex FFT in double precision (stands for numpy's pocketfft rfft,
 * spectrogram.py:170; pocketfft itself is a third-party dependency of the reference) */
static void fft_radix2(double* re, double* im, int n) {
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1;
        for (int start = 0; start < n; start += len) {
            for (int k = 0; k < half; k++) {
                double ang = -2.0*M_PI*(double)k/(double)len;
                double wr = cos(ang), wi = sin(ang);
                int a = start + k, b = a + half;
                double xr = re[b]*wr - im[b]*wi, xi = re[b]*wi + im[b]*wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
        }
    }
}

static inline float stream_sample(const float* pcm, int64_t total, int c, int64_t idx) {
    /* audio/module.py:110-111,124-129: the ring starts as zeros; stream index < 0 is still zero */
    return (idx < 0 || idx >= total) ? 0.0f : pcm[(int64_t)c*total + idx];
}

static void fft_magnitude(const float* pcm, int64_t total, int channels, int64_t tell,
                          int fft_n, int window_kind, int amplitude, float* out) {
    const int n = 1 << fft_n;
    const int bins = n/2 + 1;
    double* win = (double*)malloc(sizeof(double)*n);
    double* re = (double*)malloc(sizeof(double)*n);
    double* im = (double*)malloc(sizeof(double)*n);
    sfo_window(window_kind, n, win);
    for (int c = 0; c < channels; c++) {
        /* audio/module.py:137-138: data[:, -(n+1):-1] → stream[tell-n-1 : tell-1] */
        for (int i = 0; i < n; i++) {
            re[i] = win[i]*(double)stream_sample(pcm, total, c, tell - n - 1 + i);   /* f64 × f32 → f64 */
            im[i] = 0.0;
        }
        fft_radix2(re, im, n);
        for (int k = 0; k < bins; k++)                          /* spectrogram.py:22-26 then .astype(f32) */
            out[(int64_t)c*bins + k] = amplitude ? (float)hypot(re[k], im[k]) : (float)(re[k]*re[k] + im[k]*im[k]);
    }
    free(win); free(re); free(im);
}

void sfo_fft_power(const float* pcm, int64_t total, int channels, int64_t tell,
                   int fft_n, int window_kind, float* out) {      /* FourierMagnitude.Power, spectrogram.py:25-26 */
    fft_magnitude(pcm, total, channels, tell, fft_n, window_kind, 0, out);
}
void sfo_fft_amplitude(const float* pcm, int64_t total, int channels, int64_t tell,
                       int fft_n, int window_kind, float* out) {  /* FourierMagnitude.Amplitude = np.abs, spectrogram.py:22-23 */
    fft_magnitude(pcm, total, channels, tell, fft_n, window_kind, 1, out);
}

/* spectrogram.py:158-167 `samplerate.resample(x, ratio, 'linear')`: libsamplerate's linear converter as src_simple drives it (a
 * fresh state: last_value = in[0], last_position = 0, a constant ratio). The `samplerate` package (python-samplerate 0.x, a binding of
 * libsamplerate's src_linear.c) is NOT vendored in /root/reference and not importable in this environment: PARITY UNPINNED for this
 * option — what follows restates the published algorithm of src_linear.c (linear_vari_process: the "samples before the first sample
 * of the input array" loop, then the main loop; input_index += 1/ratio in float64; fmod_one / lrint carry the integer part into in_used),
 * sequentially, sample by sample, as the library does. The product derives read positions with its own copy of the position loop and
 * interpolates on the device; tests compare the two on random data and both with a vector computed by hand. Returns frames generated. */
static double fmod_one(double x) { double res = x - (double)lrint(x); return res < 0.0 ? res + 1.0 : res; }
int sfo_resample_linear(const float* in, int n_in, double ratio, float* out, int n_out) {
    if (n_in < 1 || !(ratio > 0.0)) return 0;
    double input_index = 0.0;
    const double last_value = in[0];
    long in_used = 0;
    int gen = 0;
    while (input_index < 1.0 && gen < n_out) {
        if ((double)in_used + (1.0 + input_index) >= (double)n_in) break;
        out[gen++] = (float)(last_value + input_index*((double)in[0] - last_value));
        input_index += 1.0/ratio;
    }
    double rem = fmod_one(input_index);
    in_used += lrint(input_index - rem);
    input_index = rem;
    while (gen < n_out && (double)in_used + input_index < (double)n_in) {
        out[gen++] = (float)((double)in[in_used - 1] + input_index*((double)in[in_used] - (double)in[in_used - 1]));
        input_index += 1.0/ratio;
        rem = fmod_one(input_index);
        in_used += lrint(input_index - rem);
        input_index = rem;
    }
    return gen;
}

/* ---------------------------------------------------------------------------------------------- */
/* Filterbank */

static double scale_fwd(int scale, double x) {               /* spectrogram.py:78-87 */
    return (scale == SFO_SCALE_MEL) ? 2595.0*log10(1.0 + x/700.0) : log(x)/log(2.0);
}
static double scale_inv(int scale, double x) {
    return (scale == SFO_SCALE_MEL) ? 700.0*(pow(10.0, x/2595.0) - 1.0) : pow(2.0, x);
}
static double interp_kernel(int interp, double x) {          /* spectrogram.py:59-70 */
    if (interp == SFO_INTERP_DIRAC) return (rint(x) == 0.0) ? 1.0 : 0.0;
    if (interp == SFO_INTERP_SINC) {
        if (x == 0.0) return 1.0;
        double y = M_PI*x;
        return fabs(sin(y)/y);
    }
    const double end = 1.2;                                   /* Euler = make_euler(end=1.2), :67 */
    double t = 2.0*x/end;
    return exp(-(t*t))/(end*sqrt(M_PI));
}

int sfo_filterbank(int scale, int interp, double fmin, double fmax, int bins, int fft_n,
                   double samplerate, int32_t* indptr, int32_t* indices, float* data, int cap) {
    const int n = 1 << fft_n;
    const int fft_bins = n/2 + 1;
    /* fft_frequencies[1] = rfftfreq(n, 1/sr)[1] = 1/(n*(1/sr)), spectrogram.py:152-153 */
    const double df = 1.0/((double)n*(1.0/samplerate));
    const double a = scale_fwd(scale, fmin), b = scale_fwd(scale, fmax);
    int nnz = 0;
    indptr[0] = 0;
    for (int r = 0; r < bins; r++) {
        /* np.linspace(a, b, bins): start + step*r, endpoint forced (spectrogram.py:186-192) */
        double lin;
        if (bins == 1) lin = a;
        else {
            double step = (b - a)/(double)(bins - 1);
            lin = (r == bins - 1) ? b : a + (double)r*step;
        }
        double index = scale_inv(scale, lin)/df;
        for (int k = 0; k < fft_bins; k++) {
            float m = (float)interp_kernel(interp, index - (double)k);      /* dtype=float32, :206-209 */
            if (fabsf(m) < 1e-5f) m = 0.0f;                                  /* :212, f32 compare */
            if (m != 0.0f) {
                if (nnz < cap) { indices[nnz] = k; data[nnz] = m; }
                nnz++;
            }
        }
        indptr[r + 1] = nnz;
    }
    return (nnz <= cap) ? nnz : -nnz;
}

/* scipy.sparse csr_matvecs: for each row, for each stored entry in index order, y[row,:] += a*x[col,:]
 * in float32 (spectrogram.py:176; scipy is a third-party dependency of the reference) */
void sfo_csr_dot(const int32_t* indptr, const int32_t* indices, const float* data, int bins,
                 const float* power, int channels, int fft_bins, float* out) {
    for (int r = 0; r < bins; r++) {
        for (int c = 0; c < channels; c++) {
            float acc = 0.0f;                         /* -ffp-contract=off: no contraction */
            for (int j = indptr[r]; j < indptr[r + 1]; j++) {
                float prod = data[j]*power[(int64_t)c*fft_bins + indices[j]];
                acc = acc + prod;
            }
            out[(int64_t)r*channels + c] = acc;
        }
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* Piano notes */

int sfo_note_of_frequency(double frequency, double tuning) {           /* piano/notes.py:74-75 */
    return (int)rint(12.0*log2(frequency/tuning) + 69.0);
}
double sfo_frequency_of_note(int note, double tuning) {                 /* piano/notes.py:58-59 */
    return tuning*pow(2.0, (double)(note - 69)/12.0);
}
void sfo_from_notes(int start_note, int end_note, int piano, int bins_in, double tuning,
                    double* fmin, double* fmax, int* bins) {            /* spectrogram.py:226-245 */
    *fmin = sfo_frequency_of_note(start_note, tuning);
    *fmax = sfo_frequency_of_note(end_note, tuning);
    if (!piano) { *bins = bins_in; return; }
    double half_semitone = pow(2.0, 0.5/12.0);
    *bins = (end_note - start_note) + 1;
    *fmin /= half_semitone;
    *fmax *= half_semitone;
}

/* ---------------------------------------------------------------------------------------------- */
/* DynamicNumber */

int sfo_dyn_coeffs(const sfo_dyn_params* p, double dt, double* k1, double* k2, double* k3) {
    const double radians = 2.0*M_PI*p->frequency;                       /* dynamics.py:187-190 */
    *k3 = (p->response*p->zeta)/(2.0*M_PI*p->frequency);                /* :182-185 */
    if (radians*dt < p->zeta) {                                         /* :231-234 */
        double a = p->zeta/(M_PI*p->frequency);                         /* :172-175 */
        double b = 1.0/(radians*radians);                               /* :177-180 */
        double c1 = a*dt, c3 = 0.5*(a + dt)*dt;
        double m = c1; if (b > m) m = b; if (c3 > m) m = c3;            /* max(k1*dt, k2, 0.5*(k1+dt)*dt) */
        *k1 = a; *k2 = m;
        return 0;
    }
    double damping = radians*sqrt(fabs(p->zeta*p->zeta - 1.0));         /* :192-195 */
    double t1 = exp(-1.0*p->zeta*radians*dt);                           /* :237-242 */
    double a1 = 2.0*t1*((p->zeta <= 1.0) ? cos(damping*dt) : cosh(damping*dt));
    double t2 = 1.0/(1.0 + t1*t1 - a1)*dt;
    *k1 = t2*(1.0 - t1*t1);
    *k2 = t2*dt;
    return 1;
}

/* float32 arrays: numpy keeps float32 when the other operand is a python float (NEP 50), i.e. the
 * python scalars dt, k1, k2, k3 are rounded to float32 first. dynamics.py:197-250 */
void sfo_dyn_step_f32(const sfo_dyn_params* p, int n, float* value, float* derivative,
                      float* previous, float* integral, const float* target, double dt) {
    if (dt == 0.0) return;                                              /* :210-211 */
    const float fdt = (float)dt;
    float worst = 0.0f;                                                 /* :222 np.abs(t - v).max() */
    for (int i = 0; i < n; i++) {
        float d = target[i] - value[i];
        float a = fabsf(d);
        if (a > worst || a != a) worst = a;
    }
    if (worst < (float)p->precision) {                                  /* :222-225 */
        if (p->integrate && integral)
            for (int i = 0; i < n; i++) { float t = value[i]*fdt; integral[i] = integral[i] + t; }
        return;
    }
    double k1d, k2d, k3d;
    sfo_dyn_coeffs(p, dt, &k1d, &k2d, &k3d);
    const float k1 = (float)k1d, k2 = (float)k2d, k3 = (float)k3d;
    for (int i = 0; i < n; i++) {
        float diff = target[i] - previous[i];
        float velocity = diff/fdt;                             /* :228 */
        previous[i] = target[i];                                        /* :229 */
        float step = derivative[i]*fdt;
        float v = value[i] + step;                             /* :245 */
        value[i] = v;
        float kv = k3*velocity;
        float s1 = target[i] + kv;
        float s2 = s1 - v;
        float kd = k1*derivative[i];
        float s3 = s2 - kd;
        float acc = s3/k2;                                     /* :246 */
        float dstep = acc*fdt;
        derivative[i] = derivative[i] + dstep;                          /* :247 */
        if (p->integrate && integral) { float t = v*fdt; integral[i] = integral[i] + t; }   /* :248-249 */
    }
}

void sfo_dyn_step_f64(const sfo_dyn_params* p, double* value, double* derivative, double* previous,
                      double* integral, double target, double dt) {
    if (dt == 0.0) return;
    if (fabs(target - *value) < p->precision) {
        if (p->integrate) *integral += (*value*dt);
        return;
    }
    double k1, k2, k3;
    sfo_dyn_coeffs(p, dt, &k1, &k2, &k3);
    double velocity = (target - *previous)/dt;
    *previous = target;
    *value += (*derivative*dt);
    double acceleration = (target + k3*velocity - *value - k1*(*derivative))/k2;
    *derivative += (acceleration*dt);
    if (p->integrate) *integral += (*value*dt);
}

/* ---------------------------------------------------------------------------------------------- */
/* numpy float32 pairwise summation (numpy/_core/src/umath/loops_utils.h.src, a third-party
 * dependency of the reference): blocks of ≤128 with 8 partial sums, halves above */
static float np_pairwise_sum_f32(const float* a, int64_t n, int64_t stride) {
    if (n < 8) {
        float res = 0.0f;
        for (int64_t i = 0; i < n; i++) res = res + a[i*stride];
        return res;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j*stride];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] = r[j] + a[(i + j)*stride];
        float s01 = r[0] + r[1], s23 = r[2] + r[3], s45 = r[4] + r[5], s67 = r[6] + r[7];
        float s0123 = s01 + s23, s4567 = s45 + s67;
        float res = s0123 + s4567;
        for (; i < n; i++) res = res + a[i*stride];
        return res;
    }
    int64_t n2 = n/2;
    n2 -= n2 % 8;
    float lo = np_pairwise_sum_f32(a, n2, stride);
    float hi = np_pairwise_sum_f32(a + n2*stride, n - n2, stride);
    return lo + hi;
}

/* waveform.py:80-87: chunks = data[:, -(chunk*points + off + 1) : -(off + 1)].reshape(c, -1, chunk);
 * reducer (:15-22) over axis 2; transposed to (points, channels) */
void sfo_waveform_row(const float* pcm, int64_t total, int channels, int64_t tell,
                      int chunk_size, int points, int reducer, float* out) {
    const int64_t offset = tell % chunk_size;                           /* waveform.py:71-73 */
    const int64_t first = tell - ((int64_t)chunk_size*points + offset + 1);
    float* tmp = (float*)malloc(sizeof(float)*chunk_size);
    for (int c = 0; c < channels; c++) {
        for (int p = 0; p < points; p++) {
            int64_t base = first + (int64_t)p*chunk_size;
            float result;
            if (reducer == SFO_REDUCER_AVERAGE) {                       /* sqrt(mean(|x|)) */
                for (int i = 0; i < chunk_size; i++) tmp[i] = fabsf(stream_sample(pcm, total, c, base + i));
                float mean = np_pairwise_sum_f32(tmp, chunk_size, 1)/(float)chunk_size;
                result = sqrtf(mean);
            } else if (reducer == SFO_REDUCER_RMS) {                    /* sqrt(sqrt(mean(x**2))*2**0.5) */
                for (int i = 0; i < chunk_size; i++) { float x = stream_sample(pcm, total, c, base + i); float q = x*x; tmp[i] = q; }
                float mean = np_pairwise_sum_f32(tmp, chunk_size, 1)/(float)chunk_size;
                float r = sqrtf(mean);
                float s = r*(float)1.4142135623730951;
                result = sqrtf(s);
            } else {                                                    /* sqrt(std(x)) */
                for (int i = 0; i < chunk_size; i++) tmp[i] = stream_sample(pcm, total, c, base + i);
                float mean = np_pairwise_sum_f32(tmp, chunk_size, 1)/(float)chunk_size;
                for (int i = 0; i < chunk_size; i++) { float d = tmp[i] - mean; float q = d*d; tmp[i] = q; }
                float var = np_pairwise_sum_f32(tmp, chunk_size, 1)/(float)chunk_size;
                float sd = sqrtf(var);
                result = sqrtf(sd);
            }
            out[(int64_t)p*channels + c] = result;
        }
    }
    free(tmp);
}

/* audio/module.py:457-458 with :74-75: volume target = 2*sqrt(mean(x²))*2**0.5 (float32),
 * std target = np.std(x) (float32), x = last 0.1 s of every channel */
void sfo_volume_std(const float* pcm, int64_t total, int channels, int64_t tell, int n,
                    float* volume_target, float* std_target) {
    const int64_t count = (int64_t)n*channels;
    float* x = (float*)malloc(sizeof(float)*count);
    float* q = (float*)malloc(sizeof(float)*count);
    for (int c = 0; c < channels; c++)
        for (int i = 0; i < n; i++)
            x[(int64_t)c*n + i] = stream_sample(pcm, total, c, tell - n - 1 + i);
    for (int64_t i = 0; i < count; i++) { float s = x[i]*x[i]; q[i] = s; }
    float mean_sq = np_pairwise_sum_f32(q, count, 1)/(float)count;
    float rms = sqrtf(mean_sq);
    float twice = 2.0f*rms;
    *volume_target = twice*(float)1.4142135623730951;                   /* python float 2**0.5 → f32 (NEP 50) */

    /* np.std on the strided (channels, n) view: per-row pairwise sums added in row order */
    float sum = 0.0f;
    for (int c = 0; c < channels; c++) sum = sum + np_pairwise_sum_f32(x + (int64_t)c*n, n, 1);
    float mean = sum/(float)count;
    for (int64_t i = 0; i < count; i++) { float d = x[i] - mean; float s = d*d; q[i] = s; }
    float var = np_pairwise_sum_f32(q, count, 1)/(float)count;
    *std_target = sqrtf(var);
    free(x); free(q);
}
