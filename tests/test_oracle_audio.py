"""
Pins the audio half of the oracle (oracle/sfo_audio.c) against the golden vectors captured from the
reference's own numpy code (tests/golden/make_golden.py). CPU only.
"""
import numpy as np
import pytest

from oracle import binding as O


def i16_to_f32(pcm):
    return (pcm.astype(np.float32)/np.float32(32768.0)).astype(np.float32)


@pytest.mark.parametrize("tag,fps,sr", [
    ("60_44100", 60.0, 44100), ("30_48000", 30.0, 48000), ("24_44100", 24.0, 44100), ("59.94_44100", 60000/1001, 44100),
])
def test_clock_and_chunks(golden, tag, fps, sr):
    g = golden("clock")
    frames = len(g[f"dt_{tag}"])
    t, dt, rdt = O.clock(fps, frames)
    # bit-exact: float64 scalar arithmetic (scheduler.py:152-173, scene.py:475-479)
    assert np.array_equal(dt, g[f"dt_{tag}"])
    assert np.array_equal(t, g[f"time_{tag}"])
    total = int(sr*(frames/fps)) + 5000
    lengths, tell = O.reader(rdt, sr, 2, total)
    assert np.array_equal(lengths, g[f"len_{tag}"])
    assert np.array_equal(tell, g[f"tell_{tag}"])


def test_reader_end_of_file(golden):
    g = golden("clock")
    rdt = np.array([0.0] + [1/60]*5)
    lengths, tell = O.reader(rdt, 44100, 2, 2000)
    want = g["len_eof"].copy()
    want[want < 0] = 0                      # StopIteration → nothing appended
    assert np.array_equal(lengths, want)
    assert tell[-1] == 2000


def test_windows(golden):
    g = golden("fft")
    assert np.allclose(O.window(0, 4096), g["window_hanning_4096"], rtol=0, atol=1e-15)
    assert np.allclose(O.window(1, 4096), g["window_hann_poisson_4096"], rtol=0, atol=1e-15)
    assert np.array_equal(O.window(2, 16), np.ones(16))


def _power_close(got, want):
    # float32 power; relative 1e-5 above the float64 FFT noise floor of the frame
    atol = 1e-12*float(want.max()) if want.max() > 0 else 1e-30
    assert np.allclose(got, want, rtol=1e-5, atol=atol), float(np.abs(got - want).max())


@pytest.mark.parametrize("name", ["silence", "sine1k", "noise", "impulse", "dc"])
def test_fft_power(golden, name):
    g = golden("fft")
    pcm = g[f"in_{name}"]
    _power_close(O.fft_power(pcm, pcm.shape[1]), g[f"power_{name}"])


@pytest.mark.parametrize("name", ["sine1k", "noise", "impulse"])
def test_fft_amplitude(golden, name):
    """FourierMagnitude.Amplitude (spectrogram.py:22-23) against vectors of the imported reference (options.npz)"""
    g, o = golden("fft"), golden("options")
    pcm = g[f"in_{name}"]
    _power_close(O.fft_power(pcm, pcm.shape[1], amplitude=True), o[f"amplitude_{name}"])
    long = g["in_noise_long"]
    _power_close(O.fft_power(long, long.shape[1], fft_n=10, amplitude=True), o["amplitude_noise_n10"])


def test_fft_sizes_and_windows(golden):
    g = golden("fft")
    pcm = g["in_noise_long"]
    for n in (8, 10, 14):
        _power_close(O.fft_power(pcm, pcm.shape[1], fft_n=n), g[f"power_noise_n{n}"])
    _power_close(O.fft_power(pcm, pcm.shape[1], window_kind=1), g["power_noise_hann_poisson"])
    _power_close(O.fft_power(pcm, pcm.shape[1], window_kind=2), g["power_noise_none"])


@pytest.mark.parametrize("tag,scale,interp,bins,fft_n,sr", [
    ("piano115", 0, 0, 115, 12, 44100), ("piano119", 0, 0, 119, 12, 44100), ("octave1000", 0, 0, 1000, 12, 44100),
    ("mel64", 1, 0, 64, 12, 44100), ("octave48_sr48k_n10", 0, 0, 48, 10, 48000), ("dirac32", 0, 1, 32, 12, 44100),
    ("sinc16_n8", 0, 2, 16, 8, 44100), ("notes200", 0, 0, 200, 12, 44100),
])
def test_filterbank(golden, tag, scale, interp, bins, fft_n, sr):
    g = golden("filterbank")
    fmin, fmax = g[f"{tag}_minmax"]
    indptr, indices, data = O.filterbank(scale, interp, fmin, fmax, bins, fft_n, sr)
    assert np.array_equal(indptr, g[f"{tag}_indptr"])
    assert np.array_equal(indices, g[f"{tag}_indices"])
    # float64 exp/log differ from numpy's by ≤ 1 ulp before the float32 cast
    assert np.allclose(data, g[f"{tag}_data"], rtol=3e-7, atol=0)


def test_from_notes(golden):
    g = golden("filterbank")
    lib = O.lib()
    got = [lib.sfo_note_of_frequency(float(f), 440.0) for f in g["note_of_freq_in"]]
    assert got == list(g["note_of_freq_out"])
    got = [lib.sfo_frequency_of_note(int(i), 440.0) for i in g["freq_of_note_in"]]
    assert np.allclose(got, g["freq_of_note_out"], rtol=1e-15)
    start, end = g["piano115_notes"]
    fmin, fmax, bins = O.from_notes(int(start), int(end), True)
    assert bins == 115
    assert np.allclose([fmin, fmax], g["piano115_minmax"], rtol=1e-15)


def test_csr_dot_matches_pipeline(golden):
    g = golden("pipeline"); f = golden("filterbank")
    pcm = i16_to_f32(g["pcm_i16"]).T.copy()
    for slot, k in enumerate(g["power_frames"]):
        power = O.fft_power(pcm, int(g["tell"][k]))
        _power_close(power, g["power"][slot])
        # on the reference's own power the sparse product is bit-exact (same order, float32)
        out = O.csr_dot(f["piano115_indptr"], f["piano115_indices"], f["piano115_data"], g["power"][slot])
        assert np.array_equal(out.reshape(2, -1), g["spec_target"][k])


@pytest.mark.parametrize("tag", ["spec"])
def test_dynamics_f32(golden, tag):
    g = golden("dynamics")
    freq, zeta, resp, integ = g[f"{tag}_params"]
    targets = g[f"{tag}_targets"]
    dyn = O.DynF32(targets[0].size, freq, zeta, resp, bool(integ))
    for k, dt in enumerate(g[f"{tag}_dts"]):
        dyn.step(targets[k], float(dt))
        assert np.array_equal(dyn.value, g[f"{tag}_values"][k].ravel()), k     # bit-exact float32
        assert np.array_equal(dyn.derivative, g[f"{tag}_derivatives"][k].ravel()), k


@pytest.mark.parametrize("tag,value0", [("volume", 0.0), ("std", 0.0), ("resp", 0.0), ("cosh", 0.0), ("idle", 0.25), ("vardt", 0.0)])
def test_dynamics_f64(golden, tag, value0):
    g = golden("dynamics")
    freq, zeta, resp, integ = g[f"{tag}_params"]
    dyn = O.DynF64(value0, freq, zeta, resp, bool(integ))
    for k, dt in enumerate(g[f"{tag}_dts"]):
        dyn.step(float(g[f"{tag}_targets"][k]), float(dt))
        assert dyn.value.value == g[f"{tag}_values"][k], k                      # bit-exact float64
        assert dyn.integral.value == g[f"{tag}_integrals"][k], k
        assert dyn.derivative.value == g[f"{tag}_derivatives"][k], k


def test_full_tape(golden):
    """Whole audio tape of the Visualizer-shaped scene, frame by frame"""
    g = golden("pipeline"); f = golden("filterbank")
    fps, sr, frames = g["meta"]; frames = int(frames); sr = int(sr)
    pcm = i16_to_f32(g["pcm_i16"]).T.copy()
    t, dt, rdt = O.clock(fps, frames)
    lengths, tell = O.reader(rdt, sr, 2, pcm.shape[1])
    assert np.array_equal(tell, g["tell"])
    assert np.array_equal(t, g["time"]) and np.array_equal(dt, g["dt"])
    volume = O.DynF64(0.0, 2, 1, 0, integrate=True)
    std = O.DynF64(0.0, 10, 1, 0)
    spec = O.DynF32(2*115, 4, 1, 0)
    for k in range(frames):
        vt, st = O.volume_std(pcm, int(tell[k]), int(0.1*sr))
        assert vt == pytest.approx(float(g["vol_target"][k]), rel=2e-6, abs=1e-12)
        assert st == pytest.approx(float(g["std_target"][k]), rel=2e-6, abs=1e-12)
        assert volume.step(vt, abs(dt[k])) == pytest.approx(float(g["vol_value"][k]), rel=1e-5, abs=1e-9)
        assert std.step(st, abs(dt[k])) == pytest.approx(float(g["std_value"][k]), rel=1e-5, abs=1e-9)
        assert volume.integral.value == pytest.approx(float(g["vol_integral"][k]), rel=1e-5, abs=1e-9)
        row = O.waveform_row(pcm, int(tell[k]), 735, 180)
        assert np.allclose(row, g["wave_row"][k], rtol=2e-6, atol=1e-9)
        power = O.fft_power(pcm, int(tell[k]))
        target = O.csr_dot(f["piano115_indptr"], f["piano115_indices"], f["piano115_data"], power)
        want_t = g["spec_target"][k].ravel()
        assert np.allclose(target.ravel(), want_t, rtol=1e-5, atol=1e-12*max(1.0, float(want_t.max())))
        column = spec.step(target.ravel(), abs(dt[k]))
        want = g["spec_value"][k].ravel()
        assert np.allclose(column, want, rtol=1e-5, atol=1e-9*max(1.0, float(np.abs(want).max()))), k


def test_waveform_reducers(golden):
    g = golden("pipeline")
    pcm = i16_to_f32(g["pcm_i16"]).T.copy()
    tell = int(g["tell"][-1])
    assert tell % 735 == 0
    assert np.allclose(O.waveform_row(pcm, tell, 735, 180, 1), g["wave_rms"], rtol=2e-6, atol=1e-9)
    assert np.allclose(O.waveform_row(pcm, tell, 735, 180, 2), g["wave_std"], rtol=3e-6, atol=1e-9)


def test_dynamics_early_out_freezes_the_whole_array(golden):
    """options.npz: a float32 (2, 24) system whose target is held until max|target - value| < 1e-6 (dynamics.py:222-225), from the
    imported reference: the oracle freezes and resumes on the same frames, bit for bit"""
    o = golden("options")
    targets = o["hold_targets"].reshape(len(o["hold_dts"]), -1)
    want = o["hold_values"].reshape(targets.shape)
    system = O.DynF32(targets.shape[1], 4, 1, 0)
    got = np.stack([system.step(targets[k], float(o["hold_dts"][k])).copy() for k in range(len(targets))])
    assert np.array_equal(got, want)
    assert np.array_equal(want[200], want[399]) and not np.array_equal(want[399], want[400])


def test_linear_resampler_restatements_agree():
    """`sample_rateio != 1` (spectrogram.py:158-167) resamples with samplerate's 'linear' converter — a package that is neither vendored
    nor importable here, so this option's parity is UNPINNED: libsamplerate's src_linear.c is restated twice — the oracle generates the
    samples sequentially as the library does, the product derives the read positions with its own copy of the position loop and
    interpolates (on the device) — and the two must agree to the bit; plus a vector small enough to follow by hand."""
    from shaderflow_amd.audio.spectrogram import linear_resample_taps
    # in = [1, 3, 7], ratio 2: x[0] while input_index < 1 (0, .5), then the one-sample delay: 1, 2 | 3, 5 — six frames
    assert O.resample_linear(np.array([1, 3, 7], np.float32), 2, 6).tolist() == [1.0, 1.0, 1.0, 2.0, 3.0, 5.0]
    rng = np.random.default_rng(1)
    x = rng.standard_normal(4096).astype(np.float32)
    for ratio in (2, 3, 4, 7):
        n_out = 4096*ratio
        want = O.resample_linear(x, ratio, n_out)
        a, b, w = linear_resample_taps(4096, ratio, n_out)
        got = (x[a].astype(np.float64) + w*(x[b].astype(np.float64) - x[a].astype(np.float64))).astype(np.float32)
        assert len(want) == n_out and np.array_equal(got, want), ratio       # integer ratios deliver exactly fft_size samples
