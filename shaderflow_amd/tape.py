"""
FrameTape: the batched export path (no reference equivalent; SURVEY.md §7 "Tape, not per-frame host round-trips").

In an export the whole audio file and the frame clock are known up front, so nothing the modules compute between
two frames needs python: for a batch of frames the device computes the STFT of every frame, the filterbank
product, the DynamicNumber recurrences (spectrogram bins in float32, volume/std in float64), the waveform rows
and the per-frame uniforms, keeps them in HBM, and the fused fragment kernel renders the batch back to back
(`sfx_tape_build` + `sfx_render_tape`). The host contributes exactly what the reference evaluates in python
scalars: the scheduler's float64 clock, the PCM chunk schedule, and the per-frame DynamicNumber coefficients
(branch selection, exp/cos — dynamics.py:231-242), rounded to float32 where numpy would round them.

A scene qualifies when it is made only of the stock modules and nothing overrides `update()`/`pipeline()`;
anything else runs through the frame loop (`ShaderScene.next`), which produces the same frames.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import TYPE_CHECKING, Optional

import numpy as np

from shaderflow_amd import _native as N
from shaderflow_amd.audio.module import ShaderAudio
from shaderflow_amd.audio.reader import chunk_schedule
from shaderflow_amd.audio.spectrogram import ShaderSpectrogram
from shaderflow_amd.audio.waveform import ShaderWaveform, WaveformReducer
from shaderflow_amd.camera import ShaderCamera
from shaderflow_amd.dynamics import ShaderDynamics, dynamics_coefficients
from shaderflow_amd.module import ShaderModule
from shaderflow_amd.piano import PianoNote
from shaderflow_amd.scheduler import freewheel_clock
from shaderflow_amd.shader import ShaderProgram
from shaderflow_amd.texture import ShaderTexture

if TYPE_CHECKING:
    from shaderflow_amd.exporting import ExportingHelper
    from shaderflow_amd.scene import ShaderScene


def _coefficients_f64(system, dts) -> np.ndarray:
    out = np.zeros(len(dts), dtype=[("dt", "f8"), ("k1", "f8"), ("k2", "f8"), ("k3", "f8")])
    for k, dt in enumerate(dts):
        dt = abs(dt)
        if dt:
            k1, k2, k3, _ = dynamics_coefficients(system.frequency, system.zeta, system.response, dt)
            out[k] = (dt, k1, k2, k3)
    return out


def _coefficients_f32(system, dts) -> np.ndarray:
    wide = _coefficients_f64(system, dts)
    out = np.zeros(len(dts), dtype=[("dt", "f4"), ("k1", "f4"), ("k2", "f4"), ("k3", "f4")])
    for name in ("dt", "k1", "k2", "k3"):
        out[name] = wide[name].astype(np.float32)       # python scalar → float32, as numpy does for a float32 array
    return out


class FrameTape:
    BATCH = 60
    SCROLL_BYTES = 8 << 30           # per-frame states of a scrolling spectrogram texture that one batch may hold

    @staticmethod
    def applicable(scene: "ShaderScene") -> bool:
        from shaderflow_amd.scene import ShaderScene
        if type(scene).update is not ShaderModule.update or type(scene).pipeline is not ShaderScene.pipeline:
            return False
        stock = (ShaderCamera, ShaderDynamics, ShaderProgram, ShaderTexture, ShaderAudio, ShaderSpectrogram, ShaderWaveform)
        audios, spectrograms, waveforms = [], [], []
        for module in scene.modules:
            if module is scene or (hasattr(module, "__eq__") and isinstance(module, ShaderScene)):
                continue
            if type(module) not in stock:
                return False
            if isinstance(module, ShaderProgram) and module not in (scene.shader, scene._final):
                return False
            if isinstance(module, ShaderAudio):
                audios.append(module)
            elif isinstance(module, ShaderSpectrogram):
                spectrograms.append(module)
            elif isinstance(module, ShaderWaveform):
                waveforms.append(module)
        if scene.shader.texture.temporal != 1 or scene.shader.texture.layers != 1:
            return False
        if not (audios or spectrograms or waveforms):
            return True                                           # clock tape: only iTime/iTau/iFrame change between frames
        if len(audios) != 1 or len(spectrograms) > 1 or len(waveforms) > 1:
            return False
        audio = audios[0]
        if audio.native is None or any(m.audio is not audio for m in (*spectrograms, *waveforms)):
            return False
        if not spectrograms:
            return True                                           # Waveform-like scenes: the tape carries a private spectrogram nobody samples
        if not spectrograms[0].device_magnitude:
            return False                                          # a `magnitude` callable of the user's own runs on the host, frame by frame (audio/spectrogram.py)
        if spectrograms[0].spectrogram_bins*audio.channels > 16384:
            return False                                          # the scan kernel walks up to 16 384 values per frame (csrc/capi_audio.hip DYNAMICS_SCAN_LIMIT)
        # a scrolling spectrogram (length > 0) keeps one state of its texture per frame of a batch in HBM (sfx_tape_desc.length_samples)
        if spectrograms[0].length_samples*spectrograms[0].spectrogram_bins*audio.channels*4*FrameTape.BATCH > FrameTape.SCROLL_BYTES:
            return False
        return True                                               # any (ssaa, subsample): fused when possible, else two passes

    def __init__(self, scene: "ShaderScene", batch: Optional[int] = None, use_mfma: Optional[bool] = None):
        self.scene = scene
        self.batch = int(batch or self.BATCH)
        # filterbank product: CSR rows in scipy's summation order (bit-exact against the reference's `M.dot`, the default) or the
        # dense banded GEMM on v_mfma_f32_32x32x2_f32 (2e-6 relative: another summation order); SHADERFLOW_FILTERBANK=mfma|csr
        if use_mfma is None:
            use_mfma = os.environ.get("SHADERFLOW_FILTERBANK", "csr").lower() == "mfma"
        self.use_mfma = bool(use_mfma)
        self.audio: Optional[ShaderAudio] = next((m for m in scene.modules if isinstance(m, ShaderAudio)), None)
        self.spectrogram: Optional[ShaderSpectrogram] = next((m for m in scene.modules if isinstance(m, ShaderSpectrogram)), None)
        self.waveform: Optional[ShaderWaveform] = next((m for m in scene.modules if isinstance(m, ShaderWaveform)), None)
        self.private_spectrogram = False
        if self.audio is not None and self.spectrogram is None:
            # A scene with audio but no spectrogram module (demo.py's Waveform, or a fragment that reads iAudioVolume only): the native
            # tape is built around a spectrogram plan, so it gets the smallest one — detached from the scene again, so that no sampler,
            # uniform or update() of it exists as far as the scene is concerned (20 µs of STFT per 60 frames buys the batched path)
            registered = len(scene.modules)
            self.spectrogram = ShaderSpectrogram(scene=scene, audio=self.audio, length=0, name="iTapePrivateSpectrogram")
            self.spectrogram.fft_n = 8
            self.spectrogram.from_notes(start=PianoNote.from_frequency(440.0), end=PianoNote.from_frequency(880.0), bins=2)
            del scene.modules[registered:]
            self.private_spectrogram = True
        self.handle: Optional[N.Handle] = None
        self.frames = 0

    # host-side schedule -----------------------------------------------------------------------------------------

    def prepare(self, frames: int) -> "FrameTape":
        """Everything python contributes, for `frames` frames from time 0"""
        scene, audio, spec = self.scene, self.audio, self.spectrogram
        self.frames = frames
        times, dts, rdts = freewheel_clock(scene.fps, frames, scene.speed)
        self.times, self.dts = times, dts
        self.release()
        if audio is None:                                    # clock tape (scenes without audio modules)
            self.clock = np.zeros(frames, dtype=[("iTime", "f4"), ("iTau", "f4"), ("iSpectrogramOffset", "f4"), ("iFrame", "i4")])
            for k, t in enumerate(times):
                self.clock[k] = (t, (t/scene.runtime) % 1.0, 0.0, round(t*scene.fps))
            handle = N.Handle()
            N.check(N.lib().sfx_clock_tape_create(scene.context.handle, self.batch, C.byref(handle)))
            self.handle = handle
            return self
        self.tell = chunk_schedule(rdts, int(audio.samplerate), audio.channels, audio._file_reader.samples.shape[0])
        self.clock = np.zeros(frames, dtype=[("iTime", "f4"), ("iTau", "f4"), ("iSpectrogramOffset", "f4"), ("iFrame", "i4")])
        width = spec.length_samples
        for k, t in enumerate(times):
            self.clock[k] = (t, (t/scene.runtime) % 1.0, ((k + 1) % width)/width, round(t*scene.fps))
        self.spec_coeff = _coefficients_f32(spec.dynamics, dts)
        self.vol_coeff = _coefficients_f64(audio.volume, dts)
        self.std_coeff = _coefficients_f64(audio.std, dts)

        reducer = self.waveform.reducer if self.waveform is not None else WaveformReducer.Average
        desc = N.TapeDesc(
            points=(self.waveform._points if self.waveform is not None else 0),
            chunk_size=(self.waveform.chunk_size if self.waveform is not None else 1),
            reducer=WaveformReducer(reducer).value,
            volume_window=int(0.1*audio.samplerate),
            use_mfma=int(self.use_mfma),
            volume_integrate=int(audio.volume.integrate), std_integrate=int(audio.std.integrate),
            length_samples=int(spec.length_samples),
            precision=float(spec.dynamics.precision),
        )
        handle = N.Handle()
        N.check(N.lib().sfx_tape_create(spec.plan(), audio.native, C.byref(desc), self.batch, C.byref(handle)))
        self.handle = handle
        return self

    def release(self) -> None:
        if self.handle is not None and self.handle.value:
            N.lib().sfx_tape_destroy(self.handle)
        self.handle = None

    def bind_static_uniforms(self) -> None:
        """Uniforms and samplers that do not change over the export (scene.py:687-703 etc.), pushed once"""
        program = self.scene.shader
        if program.program is None:
            program.compile()
        if self.spectrogram is not None and not self.private_spectrogram:
            self.spectrogram.configure_texture()             # what its first update() would do
        program.use_pipeline(program.full_pipeline())

    # device work --------------------------------------------------------------------------------------------------

    def build(self, first: int, count: int) -> None:
        """Audio state of frames [first, first+count) → tape slots [0, count). Frames must be visited in order."""
        s = slice(first, first + count)
        if self.audio is None:
            clock = np.ascontiguousarray(self.clock[s])
            N.check(N.lib().sfx_tape_build(self.handle, count, None, C.cast(clock.ctypes.data, C.POINTER(N.FrameClock)), None, None, None))
            return
        tell = np.ascontiguousarray(self.tell[s])
        clock, spec = np.ascontiguousarray(self.clock[s]), np.ascontiguousarray(self.spec_coeff[s])
        vol, std = np.ascontiguousarray(self.vol_coeff[s]), np.ascontiguousarray(self.std_coeff[s])
        N.check(N.lib().sfx_tape_build(self.handle, count, N.as_ptr(tell, C.c_int64),
                                       C.cast(clock.ctypes.data, C.POINTER(N.FrameClock)),
                                       C.cast(spec.ctypes.data, C.POINTER(N.DynCoeffF32)),
                                       C.cast(vol.ctypes.data, C.POINTER(N.DynCoeffF64)),
                                       C.cast(std.ctypes.data, C.POINTER(N.DynCoeffF64))))

    def render(self, count: int, device_out: int, first_slot: int = 0) -> None:
        """Tape slots [first_slot, first_slot+count) → `count` RGB8 frames at `device_out`"""
        scene = self.scene
        N.check(N.lib().sfx_render_tape(scene.shader.program, self.handle, first_slot, count, scene.width, scene.height,
                                        int(round(scene.ssaa*1000)), scene.subsample, C.c_void_p(device_out)))

    def read(self, what: int, count: int, first_slot: int = 0) -> np.ndarray:
        """Tape content for inspection (tests)"""
        if self.audio is None:
            bins, channels = 1, 1
        else:
            bins, channels = self.spectrogram.spectrogram_bins, self.audio.channels
        shapes = {
            N.TAPE_SPECTROGRAM: (count, bins, channels), N.TAPE_TARGETS: (count, bins, channels),
            N.TAPE_WAVEFORM: (count, self.waveform._points if self.waveform is not None else 1, channels),
            N.TAPE_UNIFORMS: (count, 8), N.TAPE_LOUDNESS: (count, 2),
            N.TAPE_SCROLL: (count, bins, self.spectrogram.length_samples if self.spectrogram is not None else 1, channels),
        }
        out = np.zeros(shapes[what], np.float32)
        N.check(N.lib().sfx_tape_read(self.handle, what, first_slot, count, out.ctypes.data, out.nbytes))
        return out

    # whole export ---------------------------------------------------------------------------------------------------

    def export(self, export: "ExportingHelper", turbo: bool = True):
        """The whole export. With an initialised torch.distributed process group (one process per GPU) the frames are sharded over
        the ranks and delivered to rank 0, which owns the sink: per-rank read-out into shared memory ("host", the default) or
        contiguous HBM-resident ranges sent over RCCL ("device") — shaderflow_amd/parallel.py."""
        from shaderflow_amd.parallel import (DeviceArray, HostDelivery, PeerWindowsUnavailable, RangeTransfer, SdmaTransfer, contiguous_device_export,
                                             interleaved_host_export, interleaved_runs, is_sharded, rank_world, shard_batches, shard_frames, shard_mode)
        scene = self.scene
        total = export.total_frames
        rank, world = rank_world()
        self.prepare(total)
        self.bind_static_uniforms()
        N.check(N.lib().sfx_tape_reset(self.handle))
        frame_bytes = scene.width*scene.height*3
        context = scene.context
        batches = shard_batches(0, total, self.batch)

        def emit_frames(pointer: int, count: int, fence: Optional[int] = None, stride: int = frame_bytes) -> None:
            export.pipe_device_frames(pointer, stride, count, turbo=turbo, fence=fence)

        mode = shard_mode() if is_sharded() else "single"
        if mode.startswith("device"):
            # device modes keep every frame of the export resident in rank 0's HBM (a 60 s 4K clip: 89.6 GB of 288). A clip that does
            # not fit falls back to host mode — for ALL ranks: rank 0 decides, everybody follows
            import torch
            import torch.distributed as dist
            verdict = [True]
            if rank == 0:
                free, _ = torch.cuda.mem_get_info(context.device)
                verdict[0] = total*export.frame_bytes + (4 << 30) < free
            dist.broadcast_object_list(verdict, src=0)
            if not verdict[0]:
                if rank == 0:
                    print(f"shaderflow_amd: {total} frames of {export.frame_bytes >> 20} MiB do not fit rank 0's free device memory: "
                          f"SHADERFLOW_SHARD={mode} falls back to host mode", flush=True)
                mode = "host"
        try:
            if mode == "single":
                buffers = [context.alloc(frame_bytes*self.batch) for _ in range(2)]
                # pixel_format "yuv420p": every batch is converted on the device right behind its render (12 bytes of 8 read and 1.5
                # written per pixel: microseconds) and the PLANAR frames are what crosses PCIe — half the bytes
                planar = [context.alloc(export.frame_bytes*self.batch) for _ in range(2)] if export.planar else None
                export._device_frames_are_planar = export.planar

                def render_batch(which: int, count: int) -> None:
                    self.render(count, buffers[which])
                    if planar is not None:
                        export.to_yuv(buffers[which], planar[which], count)
                    export.fence(which)
                try:
                    # software pipeline: batch b+1 is rendering while the frames of batch b travel to the host
                    self.build(*batches[0])
                    render_batch(0, batches[0][1])
                    for index, (first, count) in enumerate(batches):
                        if index + 1 < len(batches):
                            export.render_waits_for_last_read()          # buffers[(index+1) % 2] held batch index-1
                            self.build(*batches[index + 1])
                            render_batch((index + 1) % 2, batches[index + 1][1])
                        emit_frames((planar or buffers)[index % 2], count, fence=index % 2, stride=export.frame_bytes)
                    # leave the last frame in iFinal, where the frame loop would have left it (scene.screenshot(), scene.py:439-443)
                    context.synchronize()
                    last = context.read(buffers[(len(batches) - 1) % 2] + (batches[-1][1] - 1)*frame_bytes, frame_bytes)
                    last = last.reshape(scene.height, scene.width, 3)
                    scene._final.texture.texture.write(np.ascontiguousarray(last[::-1] if export.top_down else last))
                finally:
                    context.synchronize()
                    export._device_frames_are_planar = False
                    for pointer in buffers + (planar or []):
                        context.free(pointer)
            elif mode == "host":
                # every rank reads its own batches out over its own PCIe link into shared memory; rank 0's writer thread hands
                # them to the sink in frame order (parallel.HostDelivery): no collective on the data path. pixel_format "yuv420p":
                # the rank that rendered a batch converts it (one kernel behind the render) and the PLANAR frames travel — 12.4
                # instead of 24.9 MB per 4K frame over every link
                sink_bytes = export.frame_bytes
                # (a batch and a few: the writer takes the ranks' batches in turn, so a rank needs room for the batch it is producing while its
                # previous one is being taken — with two batches per rank an 8-rank 4K export asked /dev/shm for 24 GB)
                slots = int(os.environ.get("SHADERFLOW_SHM_SLOTS", 0)) or max(4, min(self.batch + 8, (4 << 30)//sink_bytes))
                delivery = HostDelivery(context, world, rank, sink_bytes, slots, export.fileno if rank == 0 else None,
                                        interleaved_runs(world, batches))
                buffers = [context.alloc(sink_bytes*self.batch) for _ in range(2)]
                scratch = context.alloc(frame_bytes*self.batch) if export.planar else None      # (stream-ordered: one RGB batch suffices)

                def render_sink_frames(count: int, buffer: int) -> None:
                    if scratch is None:
                        self.render(count, buffer)
                    else:
                        self.render(count, scratch)
                        export.to_yuv(scratch, buffer, count)
                try:
                    interleaved_host_export(world, rank, batches, lambda first, count, buffer: self.build(first, count), render_sink_frames, delivery, buffers)
                finally:
                    delivery.finish()
                    context.synchronize()
                    for pointer in buffers + ([scratch] if scratch is not None else []):
                        context.free(pointer)
                export.frame = total
            else:
                # the north star's layout: one contiguous frame range per rank, resident in HBM, sent to rank 0 over RCCL
                import torch
                device = torch.device("cuda", context.device)
                first, last = shard_frames(total, world, rank)
                frames_here = total if rank == 0 else (last - first)
                # what is resident, sent and handed to the sink are SINK frames: rgb24, or — pixel_format "yuv420p" — the planar frames
                # the rendering rank converts behind every batch (half the bytes over every xGMI link and over rank 0's PCIe link)
                sink_bytes = export.frame_bytes
                scratch = context.alloc(frame_bytes*self.batch) if export.planar else None
                export._device_frames_are_planar = export.planar
                window = None
                if mode == "device-sdma":
                    # peer windows: rank 0's buffer is ONE raw allocation (IPC handles name whole allocations), mapped by every rank
                    if rank == 0:
                        window = context.alloc(max(1, frames_here)*sink_bytes)
                        resident = torch.as_tensor(DeviceArray(window, max(1, frames_here)*sink_bytes), device=device)
                    else:
                        resident = torch.zeros(max(1, frames_here)*sink_bytes, dtype=torch.uint8, device=device)
                    torch.cuda.synchronize(device)
                    try:
                        transfer = SdmaTransfer(world, rank, context, sink_bytes, window, probe_at=first*sink_bytes,
                                                probe_bytes=min(4096, max(0, last - first)*sink_bytes))
                    except PeerWindowsUnavailable as unavailable:
                        # decided by every rank together (the outcomes were all-gathered): the windows do not work here, the chunks travel
                        # as RCCL point-to-point calls instead — the same resident buffers, nothing rendered yet
                        if rank == 0:
                            print(f"shaderflow_amd: peer windows unavailable ({unavailable}): SHADERFLOW_SHARD=device-sdma falls back to device mode", flush=True)
                        mode = "device"
                        transfer = RangeTransfer(world, rank, device)
                else:
                    resident = torch.zeros(max(1, frames_here)*sink_bytes, dtype=torch.uint8, device=device)
                    torch.cuda.synchronize(device)              # the fill runs on torch's stream, the renders on the context's
                    transfer = RangeTransfer(world, rank, device)

                def render(first_frame, count, view):
                    if scratch is None:
                        self.render(count, view.data_ptr())
                    else:
                        self.render(count, scratch)
                        export.to_yuv(scratch, view.data_ptr(), count)
                    context.synchronize()                       # the context's stream is not torch's: complete before the send

                def emit(view, count):
                    emit_frames(view.data_ptr(), count, stride=sink_bytes)

                # a rank that raises mid-export must not leave the others waiting for its chunks, nor rank 0's whole-export allocation
                # and the peers' mappings behind: the outcome is exchanged, then every rank tears down
                failure: Optional[BaseException] = None
                try:
                    contiguous_device_export(world, rank, total, self.batch, sink_bytes, self.build, render, emit, resident, transfer)
                    export.drain()                              # `resident` outlives the queued reads
                except BaseException as error:
                    failure = error
                    transfer.abort()
                finally:
                    export._device_frames_are_planar = False
                    if scratch is not None:
                        context.synchronize()
                        context.free(scratch)
                    try:
                        if isinstance(transfer, SdmaTransfer):
                            import torch.distributed as dist
                            transfer.close()                        # the peers' mappings of the window
                            dist.barrier(group=transfer.control)    # every peer has closed its mapping
                    finally:
                        if window is not None:                      # (also after the fall-back to RCCL: the frames still live in the raw allocation)
                            del resident
                            context.synchronize()
                            context.free(window)
                if failure is not None:
                    raise failure
                if rank != 0:
                    export.frame = total
            scene.time, scene.dt, scene.rdt = self.times[-1], self.dts[-1], self.dts[-1]      # clock of the last frame
            return export.finish()
        finally:
            context.synchronize()
            self.release()
