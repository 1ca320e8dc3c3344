"""
Parity of the HIP audio kernels (STFT, filterbank, dynamics scan, waveform, loudness, frame tape) with the oracle
and with the golden vectors captured from the reference's numpy code, through the C-ABI.
Tolerance of the path (BASELINE.json north_star): spectrogram float32 within 1e-5 relative — applied above the
float64-FFT noise floor of the frame (1e-12 of its peak power); integer/byte quantities bit-exact.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import binding as O
from shaderflow_amd import _native as N
from tests.helpers import Gpu, i16_to_f32

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def close(got, want, rtol=RTOL, floor=1e-12):
    atol = floor*max(float(np.abs(want).max()), 1e-30)
    assert np.allclose(got, want, rtol=rtol, atol=atol), float(np.abs(got - want).max())


class Audio:
    def __init__(self, gpu: Gpu, pcm: np.ndarray, samplerate=44100):
        """pcm: (samples, channels) float32"""
        self.gpu, self.lib = gpu, gpu.lib
        self.pcm = np.ascontiguousarray(pcm, np.float32)
        self.planar = np.ascontiguousarray(self.pcm.T)
        self.handle = N.Handle()
        N.check(self.lib.sfx_audio_upload(gpu.ctx.handle, N.as_ptr(self.pcm, C.c_float), self.pcm.shape[0], self.pcm.shape[1], samplerate, C.byref(self.handle)))

    def plan(self, fft_n, window, indptr, indices, data, bins):
        h = N.Handle()
        indptr, indices, data = (np.ascontiguousarray(indptr, np.int32), np.ascontiguousarray(indices, np.int32), np.ascontiguousarray(data, np.float32))
        N.check(self.lib.sfx_stft_plan(self.gpu.ctx.handle, fft_n, window, bins, self.pcm.shape[1], N.as_ptr(indptr, C.c_int32),
                                       N.as_ptr(indices, C.c_int32), N.as_ptr(data, C.c_float), C.byref(h)))
        return h

    def power(self, plan, tells, fft_n):
        tells = np.ascontiguousarray(tells, np.int64)
        out = np.zeros((len(tells), self.pcm.shape[1], (1 << fft_n)//2 + 1), np.float32)
        N.check(self.lib.sfx_stft_power(plan, self.handle, N.as_ptr(tells, C.c_int64), len(tells), N.as_ptr(out, C.c_float)))
        return out

    def targets(self, plan, tells, bins, mfma):
        tells = np.ascontiguousarray(tells, np.int64)
        out = np.zeros((len(tells), bins, self.pcm.shape[1]), np.float32)
        N.check(self.lib.sfx_spectrogram_targets(plan, self.handle, N.as_ptr(tells, C.c_int64), len(tells), int(mfma), N.as_ptr(out, C.c_float)))
        return out


@pytest.fixture()
def gpu():
    g = Gpu()
    yield g
    g.close()


def trivial_csr(bins=4):
    return np.arange(bins + 1, dtype=np.int32), np.arange(bins, dtype=np.int32), np.ones(bins, np.float32)


@pytest.mark.parametrize("name", ["silence", "sine1k", "noise", "impulse", "dc"])
def test_stft_power_against_reference_vectors(gpu, golden, name):
    g = golden("fft")
    pcm = g[f"in_{name}"]                                         # (2, 4097) planar
    audio = Audio(gpu, pcm.T)
    plan = audio.plan(12, 0, *trivial_csr(), 4)
    got = audio.power(plan, [pcm.shape[1]], 12)[0]
    close(got, g[f"power_{name}"])
    close(got, O.fft_power(pcm, pcm.shape[1]))


@pytest.mark.parametrize("name", ["sine1k", "noise", "impulse"])
def test_stft_amplitude_against_reference_vectors(gpu, golden, name):
    """FourierMagnitude.Amplitude = np.abs(rfft) (spectrogram.py:22-23) on the device vs the imported reference's vectors"""
    g, o = golden("fft"), golden("options")
    pcm = g[f"in_{name}"]
    audio = Audio(gpu, pcm.T)
    plan = audio.plan(12, 0, *trivial_csr(), 4)
    N.check(gpu.lib.sfx_stft_plan_magnitude(plan, 1))
    got = audio.power(plan, [pcm.shape[1]], 12)[0]
    close(got, o[f"amplitude_{name}"])
    close(got, O.fft_power(pcm, pcm.shape[1], amplitude=True))
    N.check(gpu.lib.sfx_stft_plan_magnitude(plan, 0))
    close(audio.power(plan, [pcm.shape[1]], 12)[0], g[f"power_{name}"])
    assert gpu.lib.sfx_stft_plan_magnitude(plan, 7) != 0


def test_stft_sizes_windows_and_start_of_stream(gpu, golden):
    g = golden("fft")
    pcm = g["in_noise_long"]
    audio = Audio(gpu, pcm.T)
    for n in (8, 10, 14):
        plan = audio.plan(n, 0, *trivial_csr(), 4)
        close(audio.power(plan, [pcm.shape[1]], n)[0], g[f"power_noise_n{n}"])
    for window, tag in ((1, "hann_poisson"), (2, "none")):
        plan = audio.plan(12, window, *trivial_csr(), 4)
        close(audio.power(plan, [pcm.shape[1]], 12)[0], g[f"power_noise_{tag}"])
    # tell = 1, 735, 1470: windows that reach before the start of the stream read zeros (audio/module.py:110-111)
    plan = audio.plan(12, 0, *trivial_csr(), 4)
    tells = [1, 735, 1470, 5000]
    got = audio.power(plan, tells, 12)
    for k, t in enumerate(tells):
        close(got[k], O.fft_power(pcm, t))


@pytest.mark.parametrize("ratio,fft_n", [(2, 11), (3, 11), (4, 12), (3, 12)])
def test_sample_rateio_resamples_before_the_transform(ratio, fft_n):
    """spectrogram.py:144-171 with `sample_rateio` = 2, 3, 4: the last 2**fft_n ring samples through samplerate's 'linear' converter
    (restated: parity unpinned, sfo_audio.c says why), `np.hanning(fft_size)` over fft_size = 2**fft_n * ratio samples, rfft in float64,
    power. Ratios 2 and 4 keep the radix-2 kernel at the larger size, 3 (6 144 / 12 288 samples) takes the float64 DFT sum; against
    numpy on the oracle's resampled data, 1e-5 relative — `fft()` and `next()` (the filterbank has fft_size/2 + 1 columns)"""
    from examples.scenes import MusicBars, make
    rng = np.random.default_rng(40 + ratio)
    pcm = (0.4*rng.standard_normal((20000, 2))).astype(np.float32)
    scene = make(MusicBars, audio=(pcm, 44100))
    scene.initialize()
    scene.audio.tell = 19000
    spectrogram = scene.spectrogram
    spectrogram.fft_n, spectrogram.sample_rateio = fft_n, ratio
    size, fft_size = 2**fft_n, 2**fft_n*ratio
    assert spectrogram.fft_size == fft_size and spectrogram.fft_bins == fft_size//2 + 1
    got = spectrogram.fft()
    data = pcm.T[:, 19000 - size - 1:19000 - 1]
    resampled = np.stack([O.resample_linear(channel, ratio, fft_size) for channel in data])
    assert resampled.shape == (2, fft_size)
    spectrum = np.fft.rfft(np.hanning(fft_size)*resampled)
    want = (spectrum*spectrum.conj()).real.astype(np.float32)
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5*float(want.max())), float(np.abs(got - want).max()/want.max())
    columns = spectrogram.next()
    assert np.allclose(columns, spectrogram.spectrogram_matrix().dot(want.T).T, rtol=2e-5, atol=2e-5*float(np.abs(columns).max()))


def test_window_function_of_the_callers_own():
    """`window` may be any callable N -> array (spectrogram.py:90-108, 155-171 multiply by what it returns, in
    float64). The host evaluates it like the reference does and the plan takes the table (sfx_stft_plan_window): the device's power
    spectrum against numpy's on the reference's own formula, 1e-5 relative"""
    from examples.scenes import MusicBars, make
    rng = np.random.default_rng(4)
    pcm = (0.4*rng.standard_normal((9000, 2))).astype(np.float32)
    scene = make(MusicBars, audio=(pcm, 44100))
    scene.initialize()
    scene.audio.tell = 8000
    spectrogram = scene.spectrogram
    spectrogram.fft_n = 11
    spectrogram.window = lambda size: np.kaiser(size, 6.0)
    got = spectrogram.fft()
    window = np.kaiser(2048, 6.0)
    data = pcm.T[:, 8000 - 2048 - 1:8000 - 1]                                # get_last_n_samples excludes the newest sample
    spectrum = np.fft.rfft(window*data)
    want = (spectrum*spectrum.conj()).real.astype(np.float32)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5*float(want.max()))
    with pytest.raises(ValueError, match="shape"):
        spectrogram.window = lambda size: np.ones(size + 1)
        spectrogram.fft()


@pytest.mark.parametrize("tag,bins", [("piano115", 115), ("octave1000", 1000), ("mel64", 64)])
def test_filterbank_csr_bit_exact_and_mfma_close(gpu, golden, tag, bins):
    f = golden("filterbank")
    rng = np.random.default_rng(1)
    pcm = (0.4*rng.standard_normal((30000, 2))).astype(np.float32)
    audio = Audio(gpu, pcm)
    plan = audio.plan(12, 0, f[f"{tag}_indptr"], f[f"{tag}_indices"], f[f"{tag}_data"], bins)
    tells = np.arange(4200, 4200 + 735*37, 735)                   # 37 frames: more than one 128-column MFMA tile with 2 channels
    power = audio.power(plan, tells, 12)
    want = np.stack([O.csr_dot(f[f"{tag}_indptr"], f[f"{tag}_indices"], f[f"{tag}_data"], p) for p in power])
    csr = audio.targets(plan, tells, bins, mfma=False)
    assert np.array_equal(csr, want)                              # scipy's order, separate multiply and add
    mfma = audio.targets(plan, tells, bins, mfma=True)
    assert np.allclose(mfma, want, rtol=2e-6, atol=1e-6*float(want.max()))


def test_waveform_and_loudness(gpu, golden):
    g = golden("pipeline")
    pcm = i16_to_f32(g["pcm_i16"])
    audio = Audio(gpu, pcm)
    tells = np.ascontiguousarray(g["tell"], np.int64)
    rows = np.zeros((len(tells), 180, 2), np.float32)
    N.check(gpu.lib.sfx_waveform_rows(audio.handle, N.as_ptr(tells, C.c_int64), len(tells), 735, 180, 0, N.as_ptr(rows, C.c_float)))
    assert np.allclose(rows, g["wave_row"], rtol=RTOL, atol=1e-9)
    for reducer, key in ((1, "wave_rms"), (2, "wave_std")):
        one = np.zeros((1, 180, 2), np.float32)
        N.check(gpu.lib.sfx_waveform_rows(audio.handle, N.as_ptr(tells[-1:], C.c_int64), 1, 735, 180, reducer, N.as_ptr(one, C.c_float)))
        assert np.allclose(one[0], g[key], rtol=RTOL, atol=1e-9)
    loud = np.zeros((len(tells), 2), np.float32)
    N.check(gpu.lib.sfx_volume_std(audio.handle, N.as_ptr(tells, C.c_int64), len(tells), 4410, N.as_ptr(loud, C.c_float)))
    assert np.allclose(loud[:, 0], g["vol_target"], rtol=RTOL, atol=1e-12)
    assert np.allclose(loud[:, 1], g["std_target"], rtol=RTOL, atol=1e-12)


@pytest.mark.parametrize("mfma", [False, True])
def test_frame_tape_against_reference_pipeline(gpu, golden, mfma):
    """The whole audio tape of the Visualizer-shaped scene, in two batches, vs the reference's frame-by-frame numpy"""
    from shaderflow_amd.dynamics import dynamics_coefficients
    g = golden("pipeline"); f = golden("filterbank")
    frames = int(g["meta"][2])
    pcm = i16_to_f32(g["pcm_i16"])
    audio = Audio(gpu, pcm)
    plan = audio.plan(12, 0, f["piano115_indptr"], f["piano115_indices"], f["piano115_data"], 115)
    desc = N.TapeDesc(points=180, chunk_size=735, reducer=0, volume_window=4410, use_mfma=int(mfma),
                      volume_integrate=1, std_integrate=0, precision=1e-6)
    tape = N.Handle()
    N.check(gpu.lib.sfx_tape_create(plan, audio.handle, C.byref(desc), 64, C.byref(tape)))

    dts = g["dt"]
    def coeffs(freq, zeta, resp, dtype):
        out = np.zeros((frames, 4), dtype)
        for k, dt in enumerate(dts):
            if dt:
                k1, k2, k3, _ = dynamics_coefficients(freq, zeta, resp, abs(float(dt)))
                out[k] = (abs(dt), k1, k2, k3)
        return out
    spec_c, vol_c, std_c = coeffs(4, 1, 0, np.float32), coeffs(2, 1, 0, np.float64), coeffs(10, 1, 0, np.float64)
    clock = np.zeros(frames, dtype=[("iTime", "f4"), ("iTau", "f4"), ("iSpectrogramOffset", "f4"), ("iFrame", "i4")])
    clock["iTime"] = g["time"]; clock["iFrame"] = np.rint(g["time"]*60)
    tells = np.ascontiguousarray(g["tell"], np.int64)

    columns, rows, uniforms = [], [], []
    for first, count in ((0, 64), (64, frames - 64)):
        s = slice(first, first + count)
        args = [np.ascontiguousarray(a[s]) for a in (tells, clock, spec_c, vol_c, std_c)]
        N.check(gpu.lib.sfx_tape_build(tape, count, N.as_ptr(args[0], C.c_int64), C.cast(args[1].ctypes.data, C.POINTER(N.FrameClock)),
                                       C.cast(args[2].ctypes.data, C.POINTER(N.DynCoeffF32)), C.cast(args[3].ctypes.data, C.POINTER(N.DynCoeffF64)),
                                       C.cast(args[4].ctypes.data, C.POINTER(N.DynCoeffF64))))
        for what, shape, sink in ((N.TAPE_SPECTROGRAM, (count, 115, 2), columns), (N.TAPE_WAVEFORM, (count, 180, 2), rows), (N.TAPE_UNIFORMS, (count, 8), uniforms)):
            out = np.zeros(shape, np.float32)
            N.check(gpu.lib.sfx_tape_read(tape, what, 0, count, out.ctypes.data, out.nbytes))
            sink.append(out)
    columns, rows, uniforms = np.concatenate(columns), np.concatenate(rows), np.concatenate(uniforms)

    want_columns = g["spec_value"].reshape(frames, 230).reshape(frames, 115, 2)      # (2,115) view of the (115,2) bytes
    peak = float(np.abs(want_columns).max())
    assert np.allclose(columns, want_columns, rtol=RTOL, atol=1e-9*peak), float(np.abs(columns - want_columns).max())
    assert np.allclose(rows, g["wave_row"], rtol=RTOL, atol=1e-9)
    assert np.allclose(uniforms[:, 0], g["time"].astype(np.float32))
    assert np.allclose(uniforms[:, 2], g["vol_value"], rtol=RTOL, atol=1e-9)
    assert np.allclose(uniforms[:, 3], g["vol_integral"], rtol=RTOL, atol=1e-9)
    assert np.allclose(uniforms[:, 4], g["std_value"], rtol=RTOL, atol=1e-9)
    assert (uniforms[:, 6].view(np.int32) == np.rint(g["time"]*60).astype(np.int32)).all()
    gpu.lib.sfx_tape_destroy(tape)


def _coeff_table(freq, zeta, resp, dts, dtype):
    from shaderflow_amd.dynamics import dynamics_coefficients
    wide = np.zeros((len(dts), 4), np.float64)
    for k, dt in enumerate(dts):
        dt = abs(float(dt))
        if dt:
            k1, k2, k3, _ = dynamics_coefficients(freq, zeta, resp, dt)
            wide[k] = (dt, k1, k2, k3)
    return np.ascontiguousarray(wide.astype(dtype))       # python scalars → float32 the way numpy rounds them for a float32 array


def test_dynamics_scan_bit_exact_on_golden_targets(gpu, golden):
    """K3 alone through sfx_dynamics_scan: the DEVICE, fed the reference's own float32 targets, reproduces the reference's
    trajectory bit for bit — including frame 0 (dt = 0) and the converged stretch where the early-out freezes the whole
    system (dynamics.py:210-211, 222-225). Run in two calls to cover the state hand-over between batches."""
    g = golden("dynamics")
    frames = len(g["spec_dts"])
    targets = np.ascontiguousarray(g["spec_targets"].reshape(frames, -1))      # (frames, 230) in the (2,115) memory order
    want = g["spec_values"].reshape(frames, -1)
    f, z, r, _ = g["spec_params"]
    coeff = _coeff_table(f, z, r, g["spec_dts"], np.float32)
    state = np.zeros(3*230, np.float32)
    got = np.zeros_like(targets)
    for first, count in ((0, 47), (47, frames - 47)):
        N.check(gpu.lib.sfx_dynamics_scan(gpu.ctx.handle, count, 230, N.as_ptr(targets[first:], C.c_float),
                                          C.cast(coeff[first:].ctypes.data, C.POINTER(N.DynCoeffF32)), np.float32(1e-6),
                                          N.as_ptr(state, C.c_float), N.as_ptr(got[first:], C.c_float)))
    assert np.array_equal(got, want)
    assert np.array_equal(state[:230], want[-1])
    assert np.array_equal(state[230:460], g["spec_derivatives"].reshape(frames, -1)[-1])
    # The early-out (dynamics.py:222-225): options.npz holds a target for 392 frames — the reference's state freezes from frame 69
    # to frame 399 (max|target - value| < 1e-6 over the WHOLE array) and moves again when the target is released. A device scan that
    # froze one frame late, or per element, would differ in the bits of every later frame.
    o = golden("options")
    hold_targets = np.ascontiguousarray(o["hold_targets"].reshape(len(o["hold_dts"]), -1))
    hold_want = o["hold_values"].reshape(hold_targets.shape)
    n = hold_targets.shape[1]
    frozen = [k for k in range(1, len(hold_want)) if k > 8 and np.array_equal(hold_want[k], hold_want[k - 1])]
    assert len(frozen) > 300 and frozen[-1] == 399, "the fixture no longer exercises the early-out"
    hold_coeff = _coeff_table(4, 1, 0, o["hold_dts"], np.float32)
    hold_state = np.zeros(3*n, np.float32)
    hold_got = np.zeros_like(hold_targets)
    for first, count in ((0, 60), (60, 60), (120, len(hold_targets) - 120)):
        N.check(gpu.lib.sfx_dynamics_scan(gpu.ctx.handle, count, n, N.as_ptr(hold_targets[first:], C.c_float),
                                          C.cast(hold_coeff[first:].ctypes.data, C.POINTER(N.DynCoeffF32)), np.float32(1e-6),
                                          N.as_ptr(hold_state, C.c_float), N.as_ptr(hold_got[first:], C.c_float)))
    assert np.array_equal(hold_got, hold_want)
    assert np.array_equal(hold_state[n:2*n], o["hold_derivatives"].reshape(hold_targets.shape)[-1])
    # and the oracle agrees (the checker is itself pinned to the same fixture)
    oracle = O.DynF32(230, 4, 1, 0)
    assert np.array_equal(np.stack([oracle.step(targets[k], float(g["spec_dts"][k])).copy() for k in range(frames)]), want)


@pytest.mark.parametrize("tag", ["volume", "std", "resp", "cosh", "idle", "vardt"])
def test_scalar_dynamics_scan_bit_exact(gpu, golden, tag):
    """The float64 scalar systems (both integrator branches, response/velocity term, integrate, early-out, variable dt)
    on the device vs the reference's trajectories: value, integral and derivative bit for bit"""
    g = golden("dynamics")
    dts = g[f"{tag}_dts"]
    frames = len(dts)
    f, z, r, integrate = g[f"{tag}_params"]
    targets = np.ascontiguousarray(g[f"{tag}_targets"].astype(np.float64).reshape(frames, 1))
    coeff = _coeff_table(f, z, r, dts, np.float64)
    start = 0.25 if tag == "idle" else 0.0
    state = np.array([[start, 0.0, start, 0.0]], np.float64)            # value, derivative, previous, integral
    out = np.zeros((frames, 1, 3), np.float64)
    N.check(gpu.lib.sfx_dynamics_scan_f64(gpu.ctx.handle, frames, 1, N.as_ptr(targets, C.c_double),
                                          C.cast(coeff.ctypes.data, C.POINTER(N.DynCoeffF64)), 1e-6, int(integrate),
                                          N.as_ptr(state, C.c_double), N.as_ptr(out, C.c_double)))
    assert np.array_equal(out[:, 0, 0], g[f"{tag}_values"])
    if integrate:
        assert np.array_equal(out[:, 0, 1], g[f"{tag}_integrals"])
    assert np.array_equal(out[:, 0, 2], g[f"{tag}_derivatives"])


@pytest.mark.parametrize("n", [2049, 2050, 4097, 9000, 16384])
def test_dynamics_scan_beyond_2048_values(gpu, n):
    """`spectrogram_bins` is whatever the user says (spectrogram.py:184): 1 025 stereo bins are 2 050 values. Up to 16 384 values the
    scan keeps 4 / 8 / 16 of them per thread of its one block; bit for bit the oracle's float32 recurrence, the whole-array early-out
    (a held target: the state must freeze for every value at once, and move again) and the hand-over between two calls included"""
    rng = np.random.default_rng(n)
    frames = 120
    dts = np.full(frames, 1/60); dts[0] = 0.0
    targets = np.abs(rng.standard_normal((frames, n))).astype(np.float32)
    targets[20:100] = targets[20]                                   # held: the system converges and freezes, then moves again
    system = O.DynF32(n, 8, 1, 0)
    want = np.stack([system.step(targets[k], float(dts[k])).copy() for k in range(frames)])
    frozen = [k for k in range(30, 100) if np.array_equal(want[k], want[k - 1])]
    assert len(frozen) > 20 and not np.array_equal(want[-1], want[99]), "the fixture no longer exercises the early-out"
    coeff = _coeff_table(8, 1, 0, dts, np.float32)
    state = np.zeros(3*n, np.float32)
    got = np.zeros_like(targets)
    for first, count in ((0, 33), (33, frames - 33)):
        N.check(gpu.lib.sfx_dynamics_scan(gpu.ctx.handle, count, n, N.as_ptr(targets[first:], C.c_float),
                                          C.cast(coeff[first:].ctypes.data, C.POINTER(N.DynCoeffF32)), np.float32(1e-6),
                                          N.as_ptr(state, C.c_float), N.as_ptr(got[first:], C.c_float)))
    assert np.array_equal(got, want), np.argwhere(got != want)[:4]
    assert gpu.lib.sfx_dynamics_scan(gpu.ctx.handle, 1, 16385, N.as_ptr(targets, C.c_float), C.cast(coeff.ctypes.data, C.POINTER(N.DynCoeffF32)),
                                     np.float32(1e-6), N.as_ptr(state, C.c_float), N.as_ptr(got, C.c_float)) == N.E_UNSUPPORTED


def test_dynamics_coefficients_match_the_oracle():
    from shaderflow_amd.dynamics import dynamics_coefficients
    for dt in (1/60, 1/30, 0.004):
        for (f, z, r) in ((4, 1, 0), (10, 1, 0), (25, 1.7, -0.5), (3, 0.4, 1.5)):
            k1, k2, k3, branch = dynamics_coefficients(f, z, r, dt)
            p = O.DynParams(f, z, r, 1e-6, 0)
            a, b, c = C.c_double(), C.c_double(), C.c_double()
            assert O.lib().sfo_dyn_coeffs(C.byref(p), dt, C.byref(a), C.byref(b), C.byref(c)) == branch
            assert (a.value, b.value, c.value) == (k1, k2, k3)


def test_magnitude_callables_of_the_users_own(gpu, golden):
    """spectrogram.py:20-41, 169-171 takes ANY callable on the complex spectrum (missing 5): the device hands the float64
    rFFT over (sfx_stft_spectrum), the host applies the callable as the reference does, the device's filterbank takes the result
    (sfx_filterbank_apply). Checked against the reference's own FourierMagnitude members in fft.npz / options.npz through callables
    that merely restate them, against numpy for one that does not, and for the tape stepping aside."""
    from shaderflow_amd import ShaderScene
    from shaderflow_amd.audio import ShaderAudio
    from shaderflow_amd.audio.spectrogram import FourierMagnitude, ShaderSpectrogram
    from shaderflow_amd.tape import FrameTape
    g, o = golden("fft"), golden("options")
    pcm = g["in_noise"]                                                 # (2, 4097) planar

    class Listener(ShaderScene):
        def build(self):
            super().build()
            self.audio = ShaderAudio(scene=self, name="iAudio")
            self.audio.load(samples=np.ascontiguousarray(pcm.T), samplerate=44100)
            self.spectrogram = ShaderSpectrogram(scene=self, audio=self.audio, length=0)

    scene = Listener()
    scene.initialize()
    spectrogram, audio = scene.spectrogram, scene.audio
    audio.tell = pcm.shape[1]
    assert FrameTape.applicable(scene)
    builtin_power, builtin_next = spectrogram.fft(), spectrogram.next().copy()
    close(builtin_power, g["power_noise"])

    spectrogram.magnitude = lambda x: (x*x.conjugate()).real            # FourierMagnitude.Power restated: not the member, so the host path runs
    assert not spectrogram.device_magnitude and not FrameTape.applicable(scene)
    spectrum = spectrogram.spectrum()
    assert spectrum.shape == (2, 2049) and spectrum.dtype == np.complex128
    window = np.hanning(4096)
    want = np.fft.rfft(window*pcm[:, -4097:-1].astype(np.float64))      # the ring excludes the newest sample (audio/module.py:137-138)
    assert np.abs(spectrum - want).max() <= 1e-9*np.abs(want).max()
    assert np.array_equal(spectrogram.fft(), builtin_power)             # the same float64 pairs, the same formula, one rounding to float32
    assert np.array_equal(spectrogram.next(), builtin_next)             # … and the same CSR product

    spectrogram.magnitude = lambda x: np.abs(x)                         # FourierMagnitude.Amplitude restated
    close(spectrogram.fft(), o["amplitude_noise"])
    spectrogram.magnitude = FourierMagnitude.Amplitude
    assert spectrogram.device_magnitude
    close(spectrogram.fft(), o["amplitude_noise"])

    spectrogram.magnitude = lambda x: np.log1p(np.abs(x.real)) + 0.25*np.abs(x.imag)      # nothing the device knows
    got = spectrogram.fft()
    ref = (np.log1p(np.abs(want.real)) + 0.25*np.abs(want.imag)).astype(np.float32)
    assert got.dtype == np.float32 and np.abs(got - ref).max() <= 1e-5*np.abs(ref).max()
    matrix = spectrogram.spectrogram_matrix()
    close(spectrogram.next(), matrix.dot(got.T).T)
    spectrogram.magnitude = lambda x: np.abs(x)[:, :7]                  # a callable that loses bins: said, not sent to the device
    with pytest.raises(ValueError, match="magnitude callable returned shape"):
        spectrogram.next()
    spectrogram.magnitude = 3
    with pytest.raises(TypeError, match="not callable"):
        spectrogram.fft()
    scene.destroy() if hasattr(scene, "destroy") else None
