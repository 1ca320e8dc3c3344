"""
The reference's frame loop (scene.py:456-479, shader.py:388-405) replayed on the ORACLE for the example scenes of
examples/basic/demo.py — what the parity tests compare the reference's own frames (tests/golden/mesa.npz) and the product's
frames with. Test infrastructure: imports oracle/, never imported by the product.
"""
from __future__ import annotations

import math

import numpy as np

from oracle import binding as O


def scene_uniforms(w, h, ssaa, fps, k, times, dts, runtime, **kw):
    """scene.py:687-703 for frame k of an export"""
    return O.default_uniforms(w, h, iTime=times[k], iTau=(times[k]/runtime) % 1.0, iDuration=runtime, iDeltatime=dts[k],
                              iSSAA=float(ssaa), iFramerate=fps, iFrame=round(times[k]*fps), **kw)


def plain_scene(fragment: str, w, h, ssaa, subsample, fps, frames, pick=None, textures=None, threads=8):
    """A scene whose only state is the clock (Basic, ShaderToy, RayMarch): frames `pick` (default all) as (n, h, w, 3)"""
    times, dts, _ = O.clock(fps, frames)
    runtime = frames/fps
    out = []
    for k in (range(frames) if pick is None else pick):
        u = scene_uniforms(w, h, ssaa, fps, k, times, dts, runtime, iSubsample=subsample)
        screen = O.render(fragment, u, textures or {}, int(w*ssaa), int(h*ssaa), threads=threads)
        out.append(O.resolve(screen, w, h, subsample, threads=threads))
    return np.stack(out)


def multishader_scene(w, h, fps, frames, pick):
    """demo.py:67-89: `child` renders first (shaders update in reverse order of creation, scene.py:468-471), the main shader adds it"""
    times, dts, _ = O.clock(fps, frames)
    out = []
    for k in pick:
        u = scene_uniforms(w, h, 1, fps, k, times, dts, frames/fps)
        child = O.render("multi_child", u, {}, w, h)
        screen = O.render("multi_main", u, {"child": O.make_texture(child, "linear", True, True)}, w, h)
        out.append(O.resolve(screen, w, h, 2))
    return np.stack(out)


def multipass_scene(background, w, h, ssaa, fps, frames, threads=8):
    times, dts, _ = O.clock(fps, frames)
    bg = O.make_texture(np.flipud(background))
    out = []
    for k in range(frames):
        u = scene_uniforms(w, h, ssaa, fps, k, times, dts, frames/fps, iLayer=0)
        layer0 = O.render("multipass", u, {"background": bg}, w*ssaa, h*ssaa, threads=threads)
        u.iLayer = 1
        layer1 = O.render("multipass", u, {"background": bg, 0: O.make_texture(layer0, "linear", False, False)}, w*ssaa, h*ssaa, threads=threads)
        out.append(O.resolve(layer1, w, h, 2, threads=threads))
    return np.stack(out)


def motionblur_scene(background, w, h, fps, frames, temporal=10, threads=8):
    """The matrix rotates after every render (shader.py:405) and iFinal reads row 0 AFTER the roll: the frame rendered
    temporal-1 frames ago, black until then (texture.py:253-256, 355-356)"""
    times, dts, _ = O.clock(fps, frames)
    bg = O.make_texture(np.flipud(background))
    zeros = np.zeros((h, w, 4), np.uint8)
    rows = [[zeros, zeros] for _ in range(temporal)]
    out = []
    for k in range(frames):
        u = scene_uniforms(w, h, 1, fps, k, times, dts, frames/fps, iLayer=0)
        u.user[0] = float(temporal)
        layer0 = O.render("motionblur", u, {"background": bg}, w, h, threads=threads)
        history = [layer0] + [rows[t][0] for t in range(1, temporal)]
        u.iLayer = 1
        layer1 = O.render("motionblur", u, {t: O.make_texture(history[t], "linear", False, False) for t in range(temporal)}, w, h, threads=threads)
        rows[0] = [layer0, layer1]
        rows = [rows[-1]] + rows[:-1]
        out.append(O.resolve(rows[0][1], w, h, 2, threads=threads))
    return np.stack(out)


def life_scene(first: np.ndarray, w, h, fps, frames, period=6, temporal=10, threads=8):
    """demo.py:223-247; `first` = the (192, 108) bool array Life.setup draws, written as the BYTES of a 192-wide float texture"""
    initial = first.astype(np.float32).reshape(108, 192, 1)
    times, dts, _ = O.clock(fps, frames)
    rows = [np.zeros((108, 192, 1), np.float32) for _ in range(temporal)]
    rows[1] = initial
    out = []
    for k in range(frames):
        u = scene_uniforms(192, 108, 1, fps, k, times, dts, frames/fps)
        u.iResolution[0], u.iResolution[1] = w, h
        u.user[0], u.user[1], u.user[2] = 192, 108, period
        rows[0] = O.render_to("life_simulation", u, {1: O.make_texture(rows[1], "nearest", True, True)}, 192, 108, 1, np.float32, threads=threads)
        rows = [rows[-1]] + rows[:-1]
        uv = scene_uniforms(w, h, 1, fps, k, times, dts, frames/fps)
        screen = O.render("life_visuals", uv, {t: O.make_texture(rows[t], "nearest", True, True) for t in range(5)}, w, h, threads=threads)
        out.append(O.resolve(screen, w, h, 2, threads=threads))
    return np.stack(out)


def dynamics_scene(background, w, h, fps, frames, pick, threads=8):
    """demo.py:114-129: a float64 DynamicNumber (frequency 4, zeta 1, response 0: ShaderDynamics' defaults) follows a square wave
    set in the scene's update(); modules update in order of creation, the scene itself first (scene.py:464-467)"""
    times, dts, _ = O.clock(fps, frames)
    bg = O.make_texture(np.flipud(background))
    system = O.DynF64(0.0, 4, 1, 0)
    out = {}
    for k in range(frames):
        target = 0.5*(1 + np.sign(np.sin(2*math.pi*times[k]*0.5)))
        system.step(float(target), abs(dts[k]))
        if k in pick:
            u = scene_uniforms(w, h, 1, fps, k, times, dts, frames/fps)
            u.user[0] = system.value.value
            out[k] = O.resolve(O.render("dynamics", u, {"background": bg}, w, h, threads=threads), w, h, 2, threads=threads)
    return np.stack([out[k] for k in pick])


def audio_scene(fragment, pcm, samplerate, background, w, h, ssaa, subsample, fps, frames, pick=None, high=14000.0,
                waveform_smooth=True, threads=8, screens=None, duration=None):
    """Visualizer / MusicBars / Waveform (demo.py:157-205): the audio tape of sfo_audio.c feeding the fragments of sfo_pixel.c.
    `high`: upper note of from_notes (14 kHz Visualizer, 18 kHz MusicBars); `screens`: a dict that receives the supersampled iScreen
    of every picked frame (the edge-aware bound of the full-size tests needs the samples); `duration`: iDuration when the export is
    longer than the `frames` replayed here"""
    planar = np.ascontiguousarray(pcm.T)
    times, dts, rdts = O.clock(fps, frames)
    runtime = frames/fps if duration is None else duration
    _, tell = O.reader(rdts, samplerate, 2, planar.shape[1])
    fmin, fmax, bins = O.from_notes(O.lib().sfo_note_of_frequency(20.0, 440.0), O.lib().sfo_note_of_frequency(high, 440.0), True)
    indptr, indices, data = O.filterbank(0, 0, fmin, fmax, bins, 12, samplerate)
    volume, std, spec = O.DynF64(0.0, 2, 1, 0, integrate=True), O.DynF64(0.0, 10, 1, 0), O.DynF32(2*bins, 4, 1, 0)
    bg = O.make_texture(np.flipud(background)) if background is not None else None
    pick = list(range(frames)) if pick is None else list(pick)
    out = {}
    for k in range(frames):
        vt, st = O.volume_std(planar, int(tell[k]), int(0.1*samplerate))
        volume.step(vt, abs(dts[k])); std.step(st, abs(dts[k]))
        target = O.csr_dot(indptr, indices, data, O.fft_power(planar, int(tell[k])))
        column = spec.step(target.ravel(), abs(dts[k])).copy()
        if k not in pick:
            continue
        row = O.waveform_row(planar, int(tell[k]), max(1, int(3*samplerate/180)), 180)
        u = O.default_uniforms(w, h, iTime=times[k], iTau=(times[k]/runtime) % 1.0, iDuration=runtime, iDeltatime=dts[k],
                               iSSAA=float(ssaa), iFramerate=fps, iFrame=round(times[k]*fps), iSubsample=subsample,
                               iAudioVolume=volume.value.value, iAudioVolumeIntegral=volume.integral.value, iAudioSTD=std.value.value,
                               iSpectrogramLength=1, iSpectrogramBins=bins, iWaveformLength=180)
        tex = {"iSpectrogram": O.make_texture(column.reshape(bins, 1, 2), "nearest", True, False),
               "iWaveform": O.make_texture(row.reshape(1, 180, 2), "linear" if waveform_smooth else "nearest", False, False)}
        if bg is not None:
            tex["background"] = bg
        screen = O.render(fragment, u, tex, int(w*ssaa), int(h*ssaa), threads=threads)
        if screens is not None:
            screens[k] = screen
        out[k] = O.resolve(screen, w, h, subsample, threads=threads)
    return np.stack([out[k] for k in pick])
