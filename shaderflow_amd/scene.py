"""
ShaderScene: the scene runtime of the headless render path.

Host mirror of the reference's shaderflow/scene.py:62-932 restricted to what an export observes: module
registration and update order (non-shaders in creation order, then ShaderPrograms in reverse, :464-471), the
time integration at the END of `next()` so that frame 0 runs with time = dt = 0 (:475-479), resolution/SSAA
plumbing (`render_resolution = int(width*ssaa)`, :372-375), the global uniforms (:687-703) and
`main(width, height, fps, ssaa, subsample, output, time, …)` (:493-639). The GL window/context is replaced by
a libshaderflow_hip context (one HIP device + stream); window events, imgui and the realtime loop are out of
scope, so `main()` without `output`/`freewheel` raises.

Two execution modes produce the same frames:
  * frame loop  — `next()` per frame exactly like the reference: modules update on the host, every shader is one
                  kernel launch (fused with the SSAA resolve when final.glsl allows it);
  * frame tape  — when nothing in the scene needs python between frames (see `tape.py`), the audio state of a
                  whole batch of frames is computed on the device and the frames are rendered back-to-back
                  with no host round trips. `main(batch=None)` picks it automatically.
"""
from __future__ import annotations

import math
import os
from collections.abc import Iterable
from pathlib import Path
from typing import Any, Optional, Union

import numpy as np
from attrs import Factory, define, field

import shaderflow_amd
from shaderflow_amd import _native as N
from shaderflow_amd.camera import ShaderCamera
from shaderflow_amd.exporting import ExportingHelper
from shaderflow_amd.ffmpeg import FFmpeg
from shaderflow_amd.message import ShaderMessage
from shaderflow_amd.module import ShaderModule, logger
from shaderflow_amd.parallel import is_sharded, rank_world
from shaderflow_amd.resolution import Resolution
from shaderflow_amd.scheduler import Scheduler
from shaderflow_amd.shader import ShaderProgram
from shaderflow_amd.variable import ShaderVariable, Uniform


@define(slots=False, eq=False)
class ShaderScene(ShaderModule):

    context: N.Context = None
    """libshaderflow_hip context (device + stream) every module of the scene allocates and launches on"""

    device: Optional[int] = None
    """HIP device index; None → LOCAL_RANK (one process per GPU) or 0"""

    quality: float = field(default=50.0, converter=float)
    modules: list = Factory(list)
    camera: ShaderCamera = None
    shader: ShaderProgram = None
    _final: ShaderProgram = None

    fuse: bool = True
    clock_loop: bool = os.environ.get("SHADERFLOW_CLOCK_LOOP", "1") != "0"
    """Scenes in which only the clock moves between frames take clockloop.ClockLoop (same frames, a fifth of the python per frame)"""
    """Shade + resolve in one kernel when final.glsl's taps stay inside the pixel's own supersamples"""

    _fused_this_frame: bool = False
    _skip_render: bool = False
    shard_warmup = "auto"                                     # plain class attribute: subclasses override it like `life_period`
    """Frames a rank renders ahead of each of its batches in a multi-GPU frame-loop export so that temporal textures hold
    their history: "auto" = 2*(deepest temporal - 1) (the history a frame samples, plus iFinal's own delay of
    temporal - 1 frames, texture.py:253-256), an int, or None for unbounded feedback (Life): render everything before."""
    _initialized: bool = False

    @property
    def fbo(self):
        return self._final.texture.fbo

    @property
    def components(self) -> int:
        return self._final.texture.components

    subsample: int = field(default=2, converter=lambda x: int(max(1, x)))
    filter_model: Optional[str] = None
    """None / "spec": float bilinear weights (the default); "llvmpipe": the 8-bit fixed-point filter of the software rasteriser the
    reference's CPU path runs on — frames within 1 LSB of its frames everywhere, at the price of the table-driven and fused kernels"""

    def initialize(self) -> None:
        if self._initialized:
            return
        self._initialized = True
        if self.context is None:
            self.context = N.Context(self.device) if self.device is not None else N.default_context()
        # how LINEAR unorm8 textures are filtered (sfx_ctx_filter_model): the scene's own wish, else SHADERFLOW_FILTER_MODEL, else the float
        # weights — said every time, because the process-wide default context outlives scenes
        self.context.filter_model(self.filter_model or os.environ.get("SHADERFLOW_FILTER_MODEL") or "spec")
        logger.info(f"Initializing scene {self.name} on {self.context.info().device_name.decode()}")

        # Default modules. The reference also registers a frametimer and a keyboard (scene.py:135-136): UI-only
        # modules without uniforms, textures or duration; omitted.
        self.camera = ShaderCamera(scene=self)

        # SSAA downsampler, then the main shader (scene.py:186-195)
        self._final = ShaderProgram(scene=self, name="iFinal")
        self._final.fragment = (shaderflow_amd.resources/"shaders"/"fragment"/"final.glsl")
        self._final.texture.components = 3
        self._final.texture.dtype = np.uint8
        self._final.texture.final = True
        self._final.texture.track = 1.0
        self.shader = ShaderProgram(scene=self, name="iScreen")
        self.shader.texture.repeat(False)
        self.shader.texture.track = 1.0
        self.build()

    def __attrs_post_init__(self) -> None:
        ShaderModule.__attrs_post_init__(self)
        self.name = (self.name or type(self).__name__)

    def destroy(self) -> None:
        for module in self.modules:
            if module is not self:
                try:
                    module.destroy()
                except Exception:
                    pass

    # time --------------------------------------------------------------------------------------------------

    time: float = field(default=0.0, converter=float)
    speed: float = field(default=1.0, converter=float)
    runtime: float = field(default=10.0, converter=float)
    fps: float = field(default=60.0, converter=float)
    dt: float = field(default=0.0, converter=float)
    rdt: float = field(default=0.0, converter=float)

    @property
    def tau(self) -> float:
        return (self.time/self.runtime) % 1.0

    @property
    def cycle(self) -> float:
        return (self.tau*math.tau)

    @property
    def frametime(self) -> float:
        return (1.0/self.fps)

    @frametime.setter
    def frametime(self, value: float):
        self.fps = (1.0/value)

    @property
    def frame(self) -> int:
        return round(self.time*self.fps)

    @frame.setter
    def frame(self, value: int):
        self.time = (value/self.fps)

    @property
    def duration(self) -> float:
        return self.runtime

    @property
    def max_duration(self) -> float:
        return max((module.duration or 0.0) for module in self.modules)

    def set_duration(self, override: Optional[float] = None) -> float:
        self.runtime = (override or self.max_duration)
        self.runtime /= self.speed
        return self.runtime

    # resolution ----------------------------------------------------------------------------------------------

    title: str = "ShaderFlow"
    _width: int = field(default=1920)
    _height: int = field(default=1080)
    _ssaa: float = field(default=1.0, converter=lambda x: max(0.01, float(x)))
    _aspect_ratio: Optional[float] = None

    @property
    def width(self) -> int:
        return self._width

    @width.setter
    def width(self, value: int):
        self.resize(width=value)

    @property
    def height(self) -> int:
        return self._height

    @height.setter
    def height(self, value: int):
        self.resize(height=value)

    @property
    def ssaa(self) -> float:
        return self._ssaa

    @ssaa.setter
    def ssaa(self, value: float):
        self._ssaa = max(0.01, float(value))
        self.relay(ShaderMessage.Shader.RecreateTextures)

    @property
    def resolution(self) -> tuple[int, int]:
        return (self.width, self.height)

    @resolution.setter
    def resolution(self, value: tuple[int, int]):
        self.resize(*value)

    @property
    def render_resolution(self) -> tuple[int, int]:
        return (int(self.width*self.ssaa), int(self.height*self.ssaa))

    @property
    def aspect_ratio(self) -> float:
        return self._aspect_ratio or (self.width/self.height)

    @aspect_ratio.setter
    def aspect_ratio(self, value: Optional[Union[float, str]]):
        if isinstance(value, str):
            text = value.replace(":", "/").strip().lower()
            if text in ("none", "false", ""):
                value = None
            else:
                num, _, den = text.partition("/")
                value = float(num)/float(den or 1)
        self._aspect_ratio = value

    def resize(self, width=None, height=None, ratio=None, bounds=None, ssaa=None, scale: float = 1.0) -> tuple[int, int]:
        self.aspect_ratio = (ratio or self._aspect_ratio)
        self._ssaa = (ssaa or self._ssaa)
        resolution = Resolution.fit(old=(self._width, self._height), new=(width, height), max=bounds,
                                    ar=self._aspect_ratio, scale=scale)
        if (resolution != (self.width, self.height)):
            self._width, self._height = resolution
            self.relay(ShaderMessage.Shader.RecreateTextures)
            logger.info(f"Resized to {self.resolution}")
        return self.resolution

    def screenshot(self) -> np.ndarray:
        """(height, width, components) uint8, top row first (scene.py:439-443)"""
        return np.flipud(self._final.texture.texture.read())

    # frame loop -------------------------------------------------------------------------------------------------

    scheduler: Scheduler = Factory(Scheduler)
    ffmpeg: FFmpeg = Factory(FFmpeg)
    """Encoder command-line builder of the export (scene.py:85): `scene.ffmpeg.h264(crf=18)`, `.h265()`, …"""
    vsync: Any = None
    quit: bool = False
    realtime: bool = True
    exporting: bool = False
    freewheel: bool = False
    headless: bool = False
    mouse_gluv: tuple = (0.0, 0.0)
    mouse_inside: bool = False
    mouse_buttons: dict = Factory(lambda: {k: False for k in range(1, 6)})

    def _can_fuse(self, program: ShaderProgram) -> bool:
        final = self._final.texture
        return bool(
            self.fuse and (program.texture.temporal == 1) and (program.texture.layers == 1)
            and (final.components == 3) and (final.dtype == np.dtype(np.uint8))
            and N.lib().sfx_fused_supported(int(round(self.ssaa*1000)), self.subsample)
            and (program.program is None or N.lib().sfx_program_fusable(program.program, int(self.ssaa)))
            and not any(isinstance(m, ShaderProgram) and m not in (program, self._final) and "iScreen" in m.fragment for m in self.modules)
        )

    def next(self, dt: float = 0.0) -> None:
        """Update every module, render every shader, then integrate time (scene.py:456-479)"""
        self._fused_this_frame = False
        for module in self.modules:
            if not isinstance(module, ShaderProgram):
                module.update()
        for module in reversed(self.modules):
            if isinstance(module, ShaderProgram):
                module.update()

        if self.vsync is not None:
            self.vsync.fps = self.fps
        self.dt = dt*self.speed
        self.rdt = dt
        self.time += self.dt

    def main(self, *,
        width: Optional[int] = 1920,
        height: Optional[int] = 1080,
        scale: float = 1.0,
        ratio: Optional[Union[float, str]] = None,
        fps: float = 60.0,
        frameskip: bool = True,
        fullscreen: bool = False,
        quality: float = 50.0,
        ssaa: float = 1.0,
        subsample: int = 2,
        output: Optional[Union[Path, str, type]] = None,
        time: Optional[float] = None,
        speed: float = 1.0,
        freewheel: bool = False,
        raw: bool = False,
        turbo: bool = True,
        buffers: int = 5,
        batch: Optional[bool] = None,
        top_down: Optional[bool] = None,
        shard: Optional[tuple[int, int]] = None,
        pixel_format: Optional[str] = None,
    ) -> Optional[Union[Path, bytes]]:
        """Render the scene to `output` (scene.py:493-639). `output` may be a path (raw rgb24 frames, or a video
        when an `ffmpeg` binary exists), "pipe"/"-"/bytes (returns the raw frames), or None with freewheel=True
        (renders without writing). `batch`: None = frame tape when the scene allows it, False = frame loop.
        `top_down`: write frame rows top-down on the device (None = exactly when an ffmpeg process is the sink).
        `shard`: (rank, world) to run ONE rank's share of a sharded frame-loop export without a process group (tests,
        external launchers); with torch.distributed initialised the group's rank and size are used.
        `pixel_format`: "rgb24" (default: the reference's stream) or "yuv420p" — converted on the device (ExportingHelper.pixel_format)."""
        self.initialize()
        self.exporting = (bool(output))
        self.freewheel = (self.exporting or freewheel)
        self.headless = (self.freewheel)
        self.realtime = (not self.headless)
        if self.realtime:
            raise NotImplementedError("Realtime windows are outside the headless render path: pass output=… or freewheel=True")
        self.title = (f"ShaderFlow • {self.name}")
        self.subsample = (subsample)
        self.quality = (quality)
        self.speed = (speed)
        self.fps = (fps)
        self.time = 0
        self.dt = 0.0
        self.rdt = 0.0
        self.relay(ShaderMessage.Shader.Compile)
        self.scheduler.clear()

        _width, _height = self.resize(width=width, height=height, ratio=ratio, scale=scale)

        for module in self.modules:
            module.setup()

        self.set_duration(eval(time) if isinstance(time, str) else time)

        if self.freewheel and (raw or self.ssaa < 1):
            self.resize(*self.render_resolution, scale=1, ssaa=1)
        else:
            self.ssaa = ssaa

        export = ExportingHelper(self, top_down=top_down, pixel_format=(pixel_format or os.environ.get("SHADERFLOW_PIXEL_FORMAT") or "rgb24"))
        if (self.exporting):
            # Every rank of a sharded export resolves the sink the same way — its kind decides the row order the kernels write
            # (exporting.py:94-118) — but only rank 0 opens it and owns the read-out ring (tape.py, _sharded_frame_loop)
            owner = (rank_world()[0] == 0)
            export.ffmpeg_clean()
            export.ffmpeg_sizes(width=_width, height=_height)
            export.ffmpeg_output(output)
            if owner:
                export.make_buffers(buffers)
            export.ffhook()
            export.popen(open_sink=owner)
        if (self.freewheel):
            export.open_bar()

        from shaderflow_amd.tape import FrameTape
        use_tape = FrameTape.applicable(self) if batch is None else bool(batch)
        if use_tape:
            return FrameTape(self).export(export, turbo=turbo)

        self.vsync = self.scheduler.new(task=self.next, frequency=self.fps, freewheel=self.freewheel,
                                        frameskip=frameskip, precise=True)
        if self.exporting and (is_sharded() or shard is not None):
            return self._sharded_frame_loop(export, turbo, *(shard or rank_world()))
        # nothing but the clock moves between frames (layered / temporal scenes without python logic): the lean loop, same frames
        from shaderflow_amd.clockloop import ClockLoop
        if self.freewheel and batch is None and self.clock_loop and ClockLoop.applicable(self):
            return ClockLoop(self).run(export, turbo)
        while (task := self.scheduler.next()):
            if (task is not self.vsync):
                continue
            if (self.quit):
                break
            export.pipe(turbo=turbo)
            export.update()
            if (export.finished):
                return export.finish()

    def _one_frame(self) -> None:
        """Run the scheduler up to and including the next vsync task (one `next`)"""
        while (task := self.scheduler.next()):
            if task is self.vsync:
                return

    def _sharded_frame_loop(self, export: ExportingHelper, turbo: bool, rank: int, world: int, batch: int = 30):
        """Multi-GPU export of a frame-loop scene (python logic between frames, layered/temporal textures): every rank steps
        through all frames so that host state stays in lock step, launches shaders only for its own batches and their
        warm-up, and rank 0 receives the finished frames in order (parallel.sharded_frame_loop)."""
        from shaderflow_amd.parallel import (FrameGather, HostDelivery, frame_modes, interleaved_host_export, interleaved_runs, shard_batches,
                                             shard_mode, sharded_frame_loop)
        total = export.total_frames
        # the frames a rank keeps, sends and delivers are SINK frames: rgb24 copies of iFinal, or — pixel_format "yuv420p" — planar frames
        # converted on the rank that rendered them, straight into the batch buffer (half the bytes over every link)
        frame_bytes = export.frame_bytes
        batches = shard_batches(0, total, batch)
        warmup = self.shard_warmup
        if warmup == "auto":
            depth = max(m.texture.temporal for m in self.modules if isinstance(m, ShaderProgram))
            warmup = 2*(depth - 1)
        modes = frame_modes(batches, world, rank, warmup)
        context = self.context
        distributed = is_sharded()

        def keep(target: int) -> None:                          # iFinal of the frame just rendered → its place in the batch buffer
            if export.planar:
                export.to_yuv(self._final.texture.texture.device_ptr(), target)
            else:
                context.copy(target, self._final.texture.texture.device_ptr(), frame_bytes)
        export._device_frames_are_planar = export.planar       # what the sink's rank pipes from device buffers is converted already
        if distributed and shard_mode() == "host":
            # every rank reads the frames of its own batches out over its own PCIe link into shared memory; rank 0's writer thread
            # hands them to the sink in frame order (parallel.HostDelivery)
            import os
            slots = int(os.environ.get("SHADERFLOW_SHM_SLOTS", 0)) or max(4, min(2*batch, (4 << 30)//frame_bytes))
            delivery = HostDelivery(context, world, rank, frame_bytes, slots, export.fileno if rank == 0 else None, interleaved_runs(world, batches))
            pointers = [context.alloc(frame_bytes*batch) for _ in range(2)]

            def advance(first: int, count: int, pointer) -> None:
                for i in range(count):
                    mode = modes[first + i]
                    self._skip_render = (mode == 0)
                    self._one_frame()
                    if mode == 2:
                        keep(pointer + i*frame_bytes)

            try:
                interleaved_host_export(world, rank, batches, advance, lambda count, pointer: None, delivery, pointers)
            finally:
                self._skip_render = False
                delivery.finish()
                context.synchronize()
                for pointer in pointers:
                    context.free(pointer)
            export.frame = total
            return export.finish()
        if distributed:
            import torch
            device = torch.device("cuda", context.device)
            tensors = [torch.zeros(frame_bytes*batch, dtype=torch.uint8, device=device) for _ in range(2)]
            torch.cuda.synchronize(device)                  # the fill runs on torch's stream, the renders on the context's
            pointer_of = lambda buffer: buffer.data_ptr()
            gather = FrameGather(world, rank, frame_bytes*batch, device)
        else:                                                   # one rank's share, no process group: frames of foreign batches are dropped
            tensors = [context.alloc(frame_bytes*batch) for _ in range(2)]
            pointer_of = lambda buffer: buffer
            gather = None

        def step(frame: int, mode: int, buffer, offset: int) -> None:
            self._skip_render = (mode == 0)
            self._one_frame()
            if mode == 2:
                keep(pointer_of(buffer) + offset)

        def emit(buffer, count: int) -> None:                   # rank 0 of a process group: frames arrive in order
            for i in range(count):
                export.pipe_device(pointer_of(buffer) + i*frame_bytes, turbo=turbo)
                export.update()
            export.drain()                                      # the gathered buffer is overwritten by a later gather

        try:
            if gather is None:
                # no process group: walk every frame, hand this rank's own batches to the sink
                for index, (first, count) in enumerate(batches):
                    for i in range(count):
                        step(first + i, modes[first + i], tensors[0], i*frame_bytes)
                    if index % world == rank:
                        context.synchronize()
                        for i in range(count):
                            export.pipe_device(tensors[0] + i*frame_bytes, turbo=False)
                            export.update()
                export.frame = total
            else:
                sharded_frame_loop(world, rank, batches, modes, step, context.synchronize, emit, gather, tensors, frame_bytes)
                if rank != 0:
                    export.frame = total
            return export.finish()
        finally:
            self._skip_render = False
            export._device_frames_are_planar = False
            context.synchronize()
            if gather is None:
                for pointer in tensors:
                    context.free(pointer)

    # module ----------------------------------------------------------------------------------------------------

    def handle(self, message) -> None:
        if isinstance(message, ShaderMessage.Window.Close):
            self.quit = True
        elif isinstance(message, (ShaderMessage.Mouse.Drag, ShaderMessage.Mouse.Position)):
            self.mouse_gluv = (message.u, message.v)

    def pipeline(self) -> Iterable[ShaderVariable]:
        yield Uniform("int", "iLayer", None)
        yield Uniform("float", "iTime", self.time)
        yield Uniform("float", "iTau", self.tau)
        yield Uniform("float", "iDuration", self.duration)
        yield Uniform("float", "iDeltatime", self.dt)
        yield Uniform("vec2", "iResolution", self.resolution)
        yield Uniform("float", "iWantAspect", self.aspect_ratio)
        yield Uniform("float", "iQuality", self.quality/100)
        yield Uniform("float", "iSSAA", self.ssaa)
        yield Uniform("float", "iFramerate", self.fps)
        yield Uniform("int", "iFrame", self.frame)
        yield Uniform("bool", "iRealtime", self.realtime)
        yield Uniform("vec2", "iMouse", self.mouse_gluv)
        yield Uniform("bool", "iMouseInside", self.mouse_inside)
        for i in range(1, 3):
            yield Uniform("bool", f"iMouse{i}", self.mouse_buttons[i])
