#!/usr/bin/env python3
"""
Golden-vector generator: runs the numpy half of the REFERENCE (read-only at /root/reference)
and stores inputs + expected outputs as small .npz fixtures in this directory.

This script only runs in the build container (it needs /root/reference). Nothing at test time
imports it; the committed .npz files are plain data (inputs and expected outputs).

What is imported from the reference (and therefore pinned by these fixtures):
  - shaderflow.scheduler.SchedulerTask            (frame clock, scheduler.py:134-173)
  - shaderflow.ffmpeg.BrokenAudioReader.stream    (PCM chunk arithmetic, ffmpeg.py:1281-1333)
  - shaderflow.audio.module.BrokenAudio           (ring buffer, audio/module.py:113-138)
  - shaderflow.audio.module.root_mean_square      (audio/module.py:74-75)
  - shaderflow.audio.spectrogram.BrokenSpectrogram(+Window/Scale/Interpolation) (spectrogram.py:44-245)
  - shaderflow.audio.waveform.WaveformReducer     (waveform.py:14-22)
  - shaderflow.dynamics.DynamicNumber             (dynamics.py:77-255)
  - shaderflow.piano.notes.PianoNote              (piano/notes.py:9-124)
  - shaderflow.resolution.Resolution              (resolution.py:6-86)

The glue between them (ShaderAudio.update audio/module.py:447-458, ShaderSpectrogram.update
spectrogram.py:298-311, ShaderWaveform.update waveform.py:80-87, ShaderScene.next scene.py:456-479)
needs a GL context in the reference, so this script restates those few lines and calls the
reference functions for all arithmetic.

Shims (capture-script only, the reference is never modified): typing.Self for python 3.10,
stub modules for GL / CLI / logging third parties that are not installed here.
"""
from __future__ import annotations

import io
import math
import sys
import types
import typing
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np

HERE = Path(__file__).resolve().parent
REFERENCE = Path("/root/reference")

# ----------------------------------------------------------------------------------------------- #
# Shims

def install_shims() -> None:
    import typing_extensions
    if not hasattr(typing, "Self"):
        typing.Self = typing_extensions.Self  # type: ignore[attr-defined]

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: (a[0] if a else None)

    dearlog = types.ModuleType("dearlog")
    dearlog.logger = _Logger()
    sys.modules["dearlog"] = dearlog

    for name in (
        "soundcard", "cyclopts", "moderngl", "_moderngl", "thefuzz", "thefuzz.process",
        "ordered_set", "watchdog", "watchdog.events", "watchdog.observers", "imgui_bundle",
        "turbopipe", "pretty_midi", "quaternion", "parsenaut", "parsenaut._cyclopts",
    ):
        if name not in sys.modules:
            sys.modules[name] = MagicMock(name=name)

    sys.path.insert(0, str(REFERENCE))


install_shims()

import shaderflow.ffmpeg as ref_ffmpeg                                    # noqa: E402
from shaderflow.audio.module import BrokenAudio, root_mean_square         # noqa: E402
from shaderflow.audio.spectrogram import (                                # noqa: E402
    BrokenSpectrogram,
    FourierMagnitude,
    SpectrogramInterpolation,
    SpectrogramScale,
    SpectrogramWindow,
)
from shaderflow.audio.waveform import WaveformReducer                     # noqa: E402
from shaderflow.dynamics import DynamicNumber                             # noqa: E402
from shaderflow.ffmpeg import BrokenAudioReader                           # noqa: E402
from shaderflow.piano.notes import PianoNote                              # noqa: E402
from shaderflow.resolution import Resolution                              # noqa: E402
from shaderflow.scheduler import SchedulerTask                            # noqa: E402

# ----------------------------------------------------------------------------------------------- #
# In-memory PCM in place of the ffmpeg decoder subprocess (ffmpeg.py:1294-1301)

class _FakeFFmpeg:
    """Stands where `FFmpeg` is looked up by BrokenAudioReader.stream; serves PCM from memory"""
    pcm_bytes: bytes = b""
    channels: int = 2
    samplerate: int = 44100

    def __getattr__(self, name):
        return lambda *a, **k: self

    def popen(self, **kwargs):
        return types.SimpleNamespace(stdout=io.BytesIO(_FakeFFmpeg.pcm_bytes))

    @staticmethod
    def get_audio_channels(path):
        return _FakeFFmpeg.channels

    @staticmethod
    def get_audio_samplerate(path):
        return _FakeFFmpeg.samplerate


RealFFmpeg = ref_ffmpeg.FFmpeg                 # the builder itself (make_golden_ffmpeg.py)
ref_ffmpeg.FFmpeg = _FakeFFmpeg


def make_reader(pcm: np.ndarray, samplerate: int) -> BrokenAudioReader:
    """pcm: (samples, channels) float32"""
    _FakeFFmpeg.pcm_bytes = np.ascontiguousarray(pcm, dtype="<f4").tobytes()
    _FakeFFmpeg.channels = pcm.shape[1]
    _FakeFFmpeg.samplerate = samplerate
    return BrokenAudioReader(path=Path("/memory.pcm"))

# ----------------------------------------------------------------------------------------------- #
# Synthetic inputs: int16-quantised so that float32 = int16/32768 is exactly reproducible

def synth_clip_i16(seconds: float, samplerate: int, seed: int) -> np.ndarray:
    """(samples, 2) int16: log sweep L / reversed R, plus a noise burst and a silent gap"""
    n = int(round(seconds*samplerate))
    t = np.arange(n, dtype=np.float64)/samplerate
    T = seconds
    k = math.log(1000.0)
    phase = 2*math.pi*20.0*T/k*(np.exp(t/T*k) - 1.0)
    left = 0.5*np.sin(phase)
    right = left[::-1].copy()
    rng = np.random.default_rng(seed)
    burst = slice(int(0.40*n), int(0.55*n))
    left[burst] += 0.35*rng.standard_normal(burst.stop - burst.start)
    right[burst] += 0.20*rng.standard_normal(burst.stop - burst.start)
    gap = slice(int(0.70*n), int(0.80*n))
    left[gap] = 0.0
    right[gap] = 0.0
    pcm = np.stack([left, right], axis=1)
    return np.clip(np.round(pcm*32767.0), -32768, 32767).astype(np.int16)


def i16_to_f32(pcm: np.ndarray) -> np.ndarray:
    return (pcm.astype(np.float32)/np.float32(32768.0)).astype(np.float32)

# ----------------------------------------------------------------------------------------------- #
# 1. Frame clock + chunk lengths

def capture_clock() -> dict:
    out = {}
    for tag, fps, samplerate, frames in (
        ("60_44100", 60.0, 44100, 3600),
        ("30_48000", 30.0, 48000, 600),
        ("24_44100", 24.0, 44100, 480),
        ("59.94_44100", 60000/1001, 44100, 600),
    ):
        dts: list[float] = []

        def tick(dt: float = 0.0):
            dts.append(dt)

        task = SchedulerTask(task=tick, frequency=fps, freewheel=True, precise=True)
        for _ in range(frames):
            task.next()
            task.fps = fps  # scene.py:476 `self.vsync.fps = self.fps`

        # scene.py:475-479: modules run with the dt/time left by the previous frame
        time_used, dt_used, rdt_used = [], [], []
        time, dt, rdt = 0.0, 0.0, 0.0
        for k in range(frames):
            time_used.append(time); dt_used.append(dt); rdt_used.append(rdt)
            dt = dts[k]*1.0
            rdt = dts[k]
            time += dt

        total = int(samplerate*(frames/fps)) + 5000
        reader = make_reader(np.zeros((total, 2), np.float32), samplerate)
        stream = reader.stream
        lengths, tell = [], 0
        tells = []
        for k in range(frames):
            reader.chunk = rdt_used[k]             # audio/module.py:450
            try:
                data = next(stream)
                tell += data.shape[0]
                lengths.append(data.shape[0])
            except StopIteration:
                lengths.append(0)
            tells.append(tell)
        out[f"dt_{tag}"] = np.array(dt_used, np.float64)
        out[f"time_{tag}"] = np.array(time_used, np.float64)
        out[f"len_{tag}"] = np.array(lengths, np.int32)
        out[f"tell_{tag}"] = np.array(tells, np.int64)

    # End-of-file behaviour: short file, reader runs dry (ffmpeg.py:1326-1327)
    reader = make_reader(np.zeros((2000, 2), np.float32), 44100)
    stream = reader.stream
    lengths = []
    for k in range(6):
        reader.chunk = 0.0 if k == 0 else 1/60
        try:
            lengths.append(next(stream).shape[0])
        except StopIteration:
            lengths.append(-1)
    out["len_eof"] = np.array(lengths, np.int32)
    return out

# ----------------------------------------------------------------------------------------------- #
# 2. FFT power (spectrogram.py:155-171) and windows (:90-108)

def make_audio(samplerate=44100) -> BrokenAudio:
    audio = BrokenAudio()
    audio._samplerate = samplerate
    audio.create_buffer()
    return audio


def capture_fft() -> dict:
    out = {}
    rng = np.random.default_rng(7)
    n = 4096
    t = np.arange(n + 1)/44100.0
    cases = {
        "silence": np.zeros((2, n + 1), np.float32),
        "sine1k": np.stack([0.5*np.sin(2*np.pi*1000*t), 0.25*np.cos(2*np.pi*3000*t)]).astype(np.float32),
        "noise": (0.3*rng.standard_normal((2, n + 1))).astype(np.float32),
        "impulse": np.eye(1, n + 1, 17, dtype=np.float32).repeat(2, 0),
        "dc": np.full((2, n + 1), 0.75, np.float32),
    }
    for name, data in cases.items():
        audio = make_audio()
        audio.add_data(data)
        spec = BrokenSpectrogram(audio=audio)
        out[f"in_{name}"] = data
        out[f"power_{name}"] = spec.fft()
    # Other sizes and windows on the noise input
    noise = (0.3*rng.standard_normal((2, 16385))).astype(np.float32)
    out["in_noise_long"] = noise
    for fft_n in (8, 10, 14):
        audio = make_audio()
        audio.add_data(noise)
        spec = BrokenSpectrogram(audio=audio, fft_n=fft_n)
        out[f"power_noise_n{fft_n}"] = spec.fft()
    for wname, window in (("hann_poisson", SpectrogramWindow.hann_poisson_window), ("none", SpectrogramWindow.none)):
        audio = make_audio()
        audio.add_data(noise)
        spec = BrokenSpectrogram(audio=audio, window=window)
        out[f"power_noise_{wname}"] = spec.fft()
    out["window_hanning_4096"] = SpectrogramWindow.hanning(4096)
    out["window_hann_poisson_4096"] = SpectrogramWindow.hann_poisson_window(4096)
    return out

# 2b. FourierMagnitude.Amplitude (spectrogram.py:22-23) on inputs of fft.npz

def capture_options() -> dict:
    out = {}
    inputs = np.load(HERE/"fft.npz")
    for name in ("sine1k", "noise", "impulse"):
        audio = make_audio()
        audio.add_data(inputs[f"in_{name}"])
        spec = BrokenSpectrogram(audio=audio, magnitude=FourierMagnitude.Amplitude)
        out[f"amplitude_{name}"] = spec.fft()
    audio = make_audio()
    audio.add_data(inputs["in_noise_long"])
    out["amplitude_noise_n10"] = BrokenSpectrogram(audio=audio, fft_n=10, magnitude=FourierMagnitude.Amplitude).fft()

    # DynamicNumber, float32 (2, 24) array like ShaderSpectrogram's (spectrogram.py:287-290): a target held long enough for the whole
    # array to come within `precision`, so that the early-out (dynamics.py:222-225) freezes value/derivative/previous, then released
    rng = np.random.default_rng(23)
    frames = 420
    targets = np.empty((frames, 2, 24), np.float32)
    targets[:8] = (np.abs(rng.standard_normal((8, 2, 24)))*3).astype(np.float32)
    targets[8:400] = targets[7]
    targets[400:] = (np.abs(rng.standard_normal((20, 2, 24)))*3).astype(np.float32)
    dts = np.full(frames, 1/60); dts[0] = 0.0
    system = DynamicNumber(frequency=4, zeta=1, response=0, dtype=np.float32)
    system.set(np.zeros((2, 24), np.float32))
    values, derivatives = [], []
    for k in range(frames):
        system.target = targets[k]
        system.next(dt=abs(float(dts[k])))
        values.append(np.array(system.value, copy=True)); derivatives.append(np.array(system.derivative, copy=True))
    out["hold_targets"], out["hold_dts"] = targets, dts
    out["hold_values"], out["hold_derivatives"] = np.asarray(values), np.asarray(derivatives)
    return out

# ----------------------------------------------------------------------------------------------- #
# 3. Filterbank CSR (spectrogram.py:186-245)

def capture_filterbank() -> dict:
    out = {}

    def dump(tag: str, spec: BrokenSpectrogram):
        m = spec.spectrogram_matrix()
        out[f"{tag}_indptr"] = m.indptr.astype(np.int32)
        out[f"{tag}_indices"] = m.indices.astype(np.int32)
        out[f"{tag}_data"] = m.data.astype(np.float32)
        out[f"{tag}_shape"] = np.array(m.shape, np.int32)
        out[f"{tag}_freqs"] = np.asarray(spec.spectrogram_frequencies, np.float64)
        out[f"{tag}_minmax"] = np.array([spec.minimum_frequency, spec.maximum_frequency], np.float64)

    spec = BrokenSpectrogram(audio=make_audio())
    spec.from_notes(start=PianoNote.from_frequency(20), end=PianoNote.from_frequency(14000), piano=True)
    dump("piano115", spec)
    out["piano115_notes"] = np.array([PianoNote.from_frequency(20).note, PianoNote.from_frequency(14000).note], np.int32)

    spec = BrokenSpectrogram(audio=make_audio())
    spec.from_notes(start=PianoNote.from_frequency(20), end=PianoNote.from_frequency(18000), piano=True)
    dump("piano119", spec)

    dump("octave1000", BrokenSpectrogram(audio=make_audio()))

    spec = BrokenSpectrogram(audio=make_audio(), scale=SpectrogramScale.MEL)
    spec.spectrogram_bins = 64
    dump("mel64", spec)

    spec = BrokenSpectrogram(audio=make_audio(48000), fft_n=10)
    spec.spectrogram_bins = 48
    dump("octave48_sr48k_n10", spec)

    spec = BrokenSpectrogram(audio=make_audio(), interpolation=SpectrogramInterpolation.Dirac)
    spec.spectrogram_bins = 32
    dump("dirac32", spec)

    spec = BrokenSpectrogram(audio=make_audio(), interpolation=SpectrogramInterpolation.Sinc, fft_n=8)
    spec.spectrogram_bins = 16
    dump("sinc16_n8", spec)

    # from_notes without piano mode (spectrogram.py:237-238)
    spec = BrokenSpectrogram(audio=make_audio())
    spec.from_notes(start="A2", end="C7", bins=200)
    dump("notes200", spec)

    # PianoNote tables (piano/notes.py:58-75)
    freqs = np.array([16.35, 20.0, 27.5, 261.63, 440.0, 1000.0, 14000.0, 18000.0, 20000.0])
    out["note_of_freq_in"] = freqs
    out["note_of_freq_out"] = np.array([PianoNote.frequency_to_index(float(f)) for f in freqs], np.int32)
    idx = np.arange(0, 140, 7)
    out["freq_of_note_in"] = idx.astype(np.int32)
    out["freq_of_note_out"] = np.array([PianoNote.index_to_frequency(int(i)) for i in idx], np.float64)
    return out

# ----------------------------------------------------------------------------------------------- #
# 4. DynamicNumber trajectories (dynamics.py:197-250)

def capture_dynamics() -> dict:
    out = {}
    rng = np.random.default_rng(11)
    frames = 90
    dts = np.full(frames, 1/60)
    dts[0] = 0.0                                     # frame 0 of a freewheel export (scene.py:475-479)

    def run(tag, system: DynamicNumber, targets, dts, f32_target=False):
        values, integrals, derivs = [], [], []
        for k in range(len(dts)):
            system.target = targets[k]
            system.next(dt=abs(float(dts[k])))
            values.append(np.array(system.value, copy=True))
            integrals.append(np.array(system.integral, copy=True))
            derivs.append(np.array(system.derivative, copy=True))
        out[f"{tag}_targets"] = np.asarray(targets)
        out[f"{tag}_dts"] = np.asarray(dts, np.float64)
        out[f"{tag}_values"] = np.asarray(values)
        out[f"{tag}_integrals"] = np.asarray(integrals)
        out[f"{tag}_derivatives"] = np.asarray(derivs)
        out[f"{tag}_params"] = np.array([system.frequency, system.zeta, system.response, float(system.integrate)])

    # (a) spectrogram-like: f32 (2,115), f=4 zeta=1 r=0 (clamped-k2 branch), spectrogram.py:287-290
    targets = (np.abs(rng.standard_normal((frames, 2, 115)))*np.linspace(0, 50, frames)[:, None, None]).astype(np.float32)
    targets[40:60] = targets[40]                      # hold → converge → early-out (dynamics.py:222-225)
    system = DynamicNumber(frequency=4, zeta=1, response=0, dtype=np.float32)
    system.set(np.zeros((2, 115), np.float32))
    run("spec", system, list(targets), dts)

    # (b) volume-like: f64 state, np.float32 targets, f=2 integrate (audio/module.py:413-417)
    tv = [np.float32(x) for x in np.abs(rng.standard_normal(frames))*0.8]
    system = DynamicNumber(value=0, frequency=2, zeta=1, response=0, integrate=True)
    system.set(system.initial, instant=True)
    run("volume", system, tv, dts)

    # (c) std-like: f=10 → pole-matching branch at 60 fps (audio/module.py:418-421)
    system = DynamicNumber(value=0, frequency=10, zeta=1, response=0)
    system.set(system.initial, instant=True)
    run("std", system, tv, dts)

    # (d) response != 0, zeta < 1 (k3 and velocity matter)
    ts = [float(np.sign(np.sin(2*np.pi*k/30))) for k in range(frames)]
    system = DynamicNumber(value=0, frequency=3, zeta=0.4, response=1.5)
    run("resp", system, ts, dts)

    # (e) zeta > 1 and very fast system → cosh pole matching
    system = DynamicNumber(value=0, frequency=25, zeta=1.7, response=-0.5)
    run("cosh", system, ts, dts)

    # (f) constant target equal to value → early-out every frame incl. integral accumulation
    system = DynamicNumber(value=0.25, frequency=2, zeta=1, response=0, integrate=True)
    run("idle", system, [0.25]*20, np.full(20, 1/60))

    # (g) variable dt (realtime-like)
    vdts = np.abs(rng.normal(1/60, 0.004, frames))
    system = DynamicNumber(value=0, frequency=4, zeta=1, response=0)
    run("vardt", system, ts, vdts)
    return out

# ----------------------------------------------------------------------------------------------- #
# 5. Whole audio tape of a Visualizer-shaped scene (demo.py:188-205)

def capture_pipeline() -> dict:
    out = {}
    fps, samplerate, frames = 60.0, 44100, 100
    pcm_i16 = synth_clip_i16(1.5, samplerate, seed=3)
    pcm = i16_to_f32(pcm_i16)
    out["pcm_i16"] = pcm_i16
    out["meta"] = np.array([fps, samplerate, frames], np.float64)

    # Frame clock exactly as scene.py:456-479 + scheduler.py:134-173
    dts: list[float] = []
    task = SchedulerTask(task=lambda dt=0.0: dts.append(dt), frequency=fps, freewheel=True, precise=True)
    for _ in range(frames):
        task.next()

    audio = make_audio(samplerate)
    reader = make_reader(pcm, samplerate)
    stream = reader.stream

    # ShaderAudio dynamics (audio/module.py:413-421); setup → reset(instant=freewheel) dynamics.py:273-274
    volume = DynamicNumber(value=0, frequency=2, zeta=1, response=0, integrate=True)
    std = DynamicNumber(value=0, frequency=10, zeta=1, response=0)
    for system in (volume, std):
        system.set(system.initial, instant=True)

    # ShaderSpectrogram (demo.py:198-203; spectrogram.py:284-290)
    spec = BrokenSpectrogram(audio=audio)
    spec.from_notes(start=PianoNote.from_frequency(20), end=PianoNote.from_frequency(14000), piano=True)
    bins = spec.spectrogram_bins
    dyn = DynamicNumber(frequency=4, zeta=1, response=0, dtype=np.float32)

    # ShaderWaveform(length=3, samplerate=60) waveform.py:31-35, 65-78
    w_length, w_rate = 3, 60
    points = w_length*w_rate
    chunk_size = max(1, int(w_length*audio.samplerate/points))

    rec = {k: [] for k in (
        "tell", "time", "dt", "vol_target", "vol_value", "vol_integral", "std_target", "std_value",
        "spec_target", "spec_value", "wave_row", "power",
    )}
    time, dt, rdt = 0.0, 0.0, 0.0
    for k in range(frames):
        # ShaderAudio.update audio/module.py:447-458
        try:
            reader.chunk = rdt
            data = next(stream).T
            audio.add_data(data)
        except StopIteration:
            pass
        volume.target = 2*root_mean_square(audio.get_last_n_seconds(0.1))*(2**0.5)
        std.target = np.std(audio.get_last_n_seconds(0.1))
        # ShaderDynamics.update dynamics.py:276-278
        volume.next(dt=abs(dt))
        std.next(dt=abs(dt))
        # ShaderWaveform.update waveform.py:80-87
        offset = audio.tell % chunk_size
        start = -int(chunk_size*points + offset + 1)
        end = -int(offset + 1)
        chunks = audio.data[:, start:end].reshape(audio.channels, -1, chunk_size)
        chunks = WaveformReducer.Average(chunks)
        wave_row = np.ascontiguousarray(chunks.T)
        # ShaderSpectrogram.update spectrogram.py:298-311
        if dyn.value.shape != (audio.channels, bins):
            dyn.set(np.zeros((audio.channels, bins), np.float32))
        power = spec.fft()
        dyn.target = spec.next().T.reshape(2, -1)
        dyn.next(dt=abs(dt))
        column = dyn.value.astype(np.float32)

        rec["tell"].append(audio.tell); rec["time"].append(time); rec["dt"].append(dt)
        rec["vol_target"].append(np.float64(volume.target)); rec["vol_value"].append(np.float64(volume.value))
        rec["vol_integral"].append(np.float64(volume.integral))
        rec["std_target"].append(np.float64(std.target)); rec["std_value"].append(np.float64(std.value))
        rec["spec_target"].append(np.array(dyn.target, copy=True)); rec["spec_value"].append(column.copy())
        rec["wave_row"].append(wave_row.astype(np.float32))
        if k in (0, 1, 2, 10, 45, 99):
            rec["power"].append(power.copy())

        # scene.py:475-479
        dt = dts[k]*1.0
        rdt = dts[k]
        time += dt

    out["power_frames"] = np.array([0, 1, 2, 10, 45, 99], np.int32)
    for key, value in rec.items():
        out[key] = np.asarray(value)
    out["bins"] = np.array([bins], np.int32)

    # Alternative reducers on one frame (waveform.py:18-22)
    blk = audio.data[:, -(chunk_size*points + 1):-1].reshape(2, -1, chunk_size)
    out["wave_rms"] = np.ascontiguousarray(WaveformReducer.RMS(blk).T).astype(np.float32)
    out["wave_std"] = np.ascontiguousarray(WaveformReducer.STD(blk).T).astype(np.float32)
    return out

# ----------------------------------------------------------------------------------------------- #
# 6. Resolution.fit (resolution.py:9-86) — the reference's only own assertions (:90-116) + extras

def capture_resolution() -> dict:
    cases = [
        dict(old=(1920, 1080)),
        dict(old=(1920, 1080), new=(1280, None)),
        dict(old=(1920, 1080), new=(None, 720)),
        dict(old=(1920, 1080), new=(1280, None), ar=16/9),
        dict(old=(1920, 1080), new=(None, 720), ar=16/9),
        dict(old=(1920, 1080), new=(1000, None), ar=2.0),
        dict(old=(1920, 1080), new=(None, 500), ar=2.0),
        dict(old=(1920, 1080), new=(1000, 720), ar=2),
        dict(old=(3840, 2160), new=(3800, 2100), max=(1920, 1080)),
        dict(old=(3000, 3000), new=(2000, 2000), max=(6000, 720), ar=16/9),
        dict(old=(1920, 1080), new=(3840, 2160), scale=0.5),
        dict(old=(1920, 1080), new=(257, 255)),
        dict(old=(1920, 1080), new=(1920, 1080), scale=1.5),
    ]
    rows = []
    for case in cases:
        w, h = Resolution.fit(**case)
        old = case.get("old", (None, None)); new = case.get("new", (None, None)); mx = case.get("max", (None, None))
        rows.append([
            *(float("nan") if v is None else float(v) for v in (*old, *new, *mx)),
            float("nan") if case.get("ar") is None else float(case["ar"]), float(case.get("scale", 1.0)), float(w), float(h),
        ])
    return {"cases": np.array(rows, np.float64)}

# ----------------------------------------------------------------------------------------------- #

def main() -> None:
    for name, fn in (
        ("clock", capture_clock),
        ("fft", capture_fft),
        ("options", capture_options),
        ("filterbank", capture_filterbank),
        ("dynamics", capture_dynamics),
        ("pipeline", capture_pipeline),
        ("resolution", capture_resolution),
    ):
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue                                                        # `make_golden.py options` regenerates one fixture
        data = fn()
        path = HERE/f"{name}.npz"
        np.savez_compressed(path, **data)
        print(f"{name:12s} {len(data):3d} arrays  {path.stat().st_size/1024:8.1f} KiB")


if __name__ == "__main__":
    main()
