"""
ExportingHelper: frame counter, read-out ring and the encoder hand-off of an export.

Host mirror of the reference's shaderflow/exporting.py:30-203. The reference reads the final FBO into one of N
GL buffers (`fbo.read_into`) and lets turbopipe write it to ffmpeg's stdin from a worker thread; here the final
RGB8 texture is copied to one of N PINNED host buffers on a copy stream (`sfx_ring_read_async`) and a native
writer thread streams it to a file descriptor (`sfx_ring_pipe`), with the same reuse fence (`turbopipe.sync` →
`sfx_ring_pipe_sync`). Frames leave exactly as the reference hands them to ffmpeg: rgb24, rows BOTTOM-UP
(ffmpeg then applies `vflip`, exporting.py:94-103).

Sinks: a path ending in .rgb/.raw (or any path when no `ffmpeg` binary exists) receives the raw frames;
"pipe"/"-"/bytes returns them; with an `ffmpeg` binary on PATH other paths are encoded by the command line
`scene.ffmpeg` (shaderflow_amd/ffmpeg.py) builds exactly as the reference does: rawvideo rgb24 stdin, scale, vflip,
the configured codecs, `module.ffhook` additions such as the audio track (exporting.py:91-120).

Encoder hand-off on the device: when frames go to an ffmpeg process the resolve kernels write the
rows top-down (`sfx_ctx_output_top_down`) and the `vflip` filter is left out of the command, which takes a full
pass over every frame off the encoder's CPU threads. Raw sinks keep the reference's byte stream (rows bottom-up)
unless `top_down=True` is asked for.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile
import time
from pathlib import Path
from typing import TYPE_CHECKING, Any, Callable, Optional

import numpy as np
from attrs import Factory, define

from shaderflow_amd import _native as N
from shaderflow_amd.module import logger

if TYPE_CHECKING:
    from shaderflow_amd.scene import ShaderScene


@define(slots=False, eq=False)
class ExportingHelper:
    scene: "ShaderScene"

    frame: int = 0
    start: float = Factory(time.monotonic)
    relay: Optional[Callable[[int, int], None]] = None
    took: Optional[float] = None

    # sink
    kind: Optional[str] = None            # "path-raw", "path-ffmpeg", "pipe"
    path: Optional[Path] = None
    process: Optional[subprocess.Popen] = None
    file: Any = None
    fileno: Optional[int] = None
    top_down: Optional[bool] = None       # None: top-down exactly when an ffmpeg process is the sink
    pixel_format: str = "rgb24"
    """What the sink receives: "rgb24" — the reference's byte stream (exporting.py:97) — or "yuv420p": planar 4:2:0 made on the device
    (sfx_rgb_to_yuv420: BT.601 limited range, or `yuv_matrix = "bt709"`), half the bytes over PCIe and the pipe; ffmpeg takes it as
    `-pix_fmt yuv420p` rawvideo and the codec's own conversion falls away. Opt-in: `scene.main(pixel_format="yuv420p")`."""
    yuv_matrix: str = "bt601"
    _yuv_slots: list = Factory(list)      # device staging of the frame loop's converted frames, one per ring slot

    # ring
    ring: Optional[N.Handle] = None
    slots: int = 0

    @property
    def total_frames(self) -> int:
        return max(1, round(self.scene.runtime*self.scene.fps))

    @property
    def planar(self) -> bool:
        return self.pixel_format == "yuv420p"

    @property
    def frame_bytes(self) -> int:
        """Bytes of one frame as the sink receives it"""
        pixels = self.scene.width*self.scene.height
        return pixels*3//2 if self.planar else pixels*3

    def to_yuv(self, rgb: int, yuv: int, frames: int = 1) -> None:
        self.scene.context.rgb_to_yuv420(rgb, yuv, self.scene.width, self.scene.height, frames, 1 if self.yuv_matrix == "bt709" else 0)

    @property
    def finished(self) -> bool:
        return (self.frame >= self.total_frames)

    def open_bar(self) -> None:
        self.start = time.monotonic()

    def update(self) -> None:
        if self.relay:
            self.relay(self.frame, self.total_frames)
        self.frame += 1

    # configuration --------------------------------------------------------------------------------------------

    @property
    def ffmpeg(self):
        return self.scene.ffmpeg

    def ffmpeg_clean(self) -> None:
        self.ffmpeg.clear(video_codec=False, audio_codec=False)

    def ffmpeg_sizes(self, width: int, height: int) -> None:
        self.ffmpeg.time = self.scene.runtime
        if self.pixel_format not in ("rgb24", "yuv420p"):
            raise ValueError(f"pixel_format {self.pixel_format!r}: 'rgb24' (the reference's stream) or 'yuv420p' (converted on the device)")
        if self.planar and (self.scene.width % 2 or self.scene.height % 2):
            raise ValueError(f"yuv420p needs even extents, the scene is {self.scene.width}x{self.scene.height}")
        self.ffmpeg.pipe_input(pixel_format=self.pixel_format, width=self.scene.width, height=self.scene.height, framerate=self.scene.fps)
        self.ffmpeg.scale(width=width, height=height)
        self.ffmpeg.vflip()

    def ffmpeg_output(self, output) -> None:
        if (output in ("pipe", "-", bytes)):
            self.kind = "pipe"
            self.ffmpeg.pipe_output()
        elif ("tcp://" in str(output)):
            raise NotImplementedError
        else:
            self.path = Path(output).expanduser().absolute()
            self.path.parent.mkdir(parents=True, exist_ok=True)
            raw = self.path.suffix.lower() in (".rgb", ".raw", ".rgb24", ".yuv", ".i420")
            self.kind = "path-raw" if (raw or not self.ffmpeg.available()) else "path-ffmpeg"
            if self.kind == "path-raw" and not raw:
                logger.warning(f"No ffmpeg binary on PATH: writing raw rgb24 frames (rows bottom-up) to {self.path}")
            self.ffmpeg.output(path=self.path)
        if self.top_down is None:
            self.top_down = (self.kind == "path-ffmpeg")
        if self.top_down:                                     # the flip happens while the frame is written on the device
            self.ffmpeg.filters = [stage for stage in self.ffmpeg.filters if stage.kind != "vflip"]
            self.ffmpeg.vflip(device=True)

    def ffhook(self) -> None:
        for module in self.scene.modules:
            module.ffhook(self.ffmpeg)

    def popen(self, open_sink: bool = True) -> None:
        """`open_sink=False`: a rank of a sharded export that renders frames but does not own the sink — it still has to write
        its rows in the order the sink wants"""
        self.scene.context.output_top_down(bool(self.top_down))
        if not open_sink:
            return
        if self.kind == "path-ffmpeg":
            # the encoder's own output goes to temporary files, never to a pipe nobody drains (the reference hands ffmpeg
            # file-backed pipes for the same reason, exporting.py:131-132): a chatty encoder cannot stall the export
            self._encoder_log = tempfile.TemporaryFile(mode="w+b")
            self.process = self.ffmpeg.popen(stdin=subprocess.PIPE, stdout=self._encoder_log, stderr=self._encoder_log)
            self.fileno = self.process.stdin.fileno()
        elif self.kind == "path-raw":
            self.file = open(self.path, "wb")
            self.fileno = self.file.fileno()
        elif self.kind == "pipe":
            self.file = tempfile.TemporaryFile(mode="w+b")
            self.fileno = self.file.fileno()

    # buffers and piping ---------------------------------------------------------------------------------------

    def make_buffers(self, n: int = 2) -> None:
        self.release_buffers()
        handle = N.Handle()
        N.check(N.lib().sfx_ring_create(self.scene.context.handle, self.frame_bytes, max(1, n), C.byref(handle)))
        self.ring, self.slots = handle, max(1, n)
        if self.planar:
            self._yuv_slots = [self.scene.context.alloc(self.frame_bytes) for _ in range(self.slots)]

    def release_buffers(self) -> None:
        if self.ring is not None and self.ring.value:
            N.check(N.lib().sfx_ring_pipe_sync(self.ring, -1))
            N.lib().sfx_ring_destroy(self.ring)
        for pointer in self._yuv_slots:
            self.scene.context.free(pointer)
        self._yuv_slots = []
        self.ring, self.slots = None, 0

    def _check_encoder(self) -> None:
        if (self.process is not None) and (self.process.poll() is not None):
            raise RuntimeError("FFmpeg process closed unexpectedly with traceback:\n" + self.encoder_output())

    def encoder_output(self) -> str:
        """What the encoder process has printed so far (stdout and stderr share one temporary file)"""
        log = getattr(self, "_encoder_log", None)
        if log is None:
            return ""
        log.flush()
        log.seek(0)
        return log.read().decode("utf-8", "replace")

    def pipe(self, turbo: bool = False) -> None:
        """Queue the frame that was just rendered (exporting.py:151-174)"""
        if (self.fileno is None) or (self.ring is None):
            return
        self._check_encoder()
        slot = self.frame % self.slots
        if self.planar:
            # the slot's last frame has left its staging buffer; then convert (on the render stream, in order with the draws) and read out
            N.check(N.lib().sfx_ring_pipe_sync(self.ring, slot))
            self.to_yuv(self.scene._final.texture.texture.device_ptr(), self._yuv_slots[slot])
            N.check(N.lib().sfx_ring_read_device_async(self.ring, C.c_void_p(self._yuv_slots[slot]), slot))
        else:
            N.check(N.lib().sfx_ring_read_async(self.ring, self.scene._final.texture.texture.handle, slot))
        N.check(N.lib().sfx_ring_pipe(self.ring, slot, self.fileno))
        if not turbo:
            N.check(N.lib().sfx_ring_pipe_sync(self.ring, slot))

    def fence(self, which: int) -> None:
        """Everything launched on the render stream so far is what frames read against fence `which` depend on"""
        if self.ring is not None:
            N.check(N.lib().sfx_ring_fence(self.ring, which))

    def render_waits_for_last_read(self) -> None:
        """The render stream may not overwrite a frame buffer before the last queued read of it has finished"""
        if self.ring is not None and self.frame > 0:
            N.check(N.lib().sfx_ring_stream_wait(self.ring, (self.frame - 1) % self.slots))

    def drain(self) -> None:
        """Every queued frame has left its device buffer and reached the sink (gathered buffers are reused right after)"""
        if self.ring is not None and self.ring.value:
            N.check(N.lib().sfx_ring_pipe_sync(self.ring, -1))

    def pipe_device(self, device_ptr: int, turbo: bool = True, fence: Optional[int] = None) -> None:
        """Same, for a frame that lives in a raw device buffer (frame tape batches)"""
        if (self.fileno is None) or (self.ring is None):
            return
        self._check_encoder()
        slot = self.frame % self.slots
        if self.planar and not getattr(self, "_device_frames_are_planar", False):
            N.check(N.lib().sfx_ring_pipe_sync(self.ring, slot))     # (an RGB frame in a device buffer: the sharded frame loop's path)
            self.to_yuv(device_ptr, self._yuv_slots[slot])
            device_ptr, fence = self._yuv_slots[slot], None
        if fence is None:
            N.check(N.lib().sfx_ring_read_device_async(self.ring, C.c_void_p(device_ptr), slot))
        else:
            N.check(N.lib().sfx_ring_read_fenced_async(self.ring, C.c_void_p(device_ptr), slot, fence))
        N.check(N.lib().sfx_ring_pipe(self.ring, slot, self.fileno))
        if not turbo:
            N.check(N.lib().sfx_ring_pipe_sync(self.ring, slot))

    def pipe_device_frames(self, device_ptr: int, stride: int, count: int, turbo: bool = True, fence: Optional[int] = None) -> None:
        """`count` consecutive frames of a batch in one native call (the per-frame python loop costs more than the frames themselves
        when they are small); with a progress relay, or without turbo, frame by frame as before"""
        if self.relay is not None or not turbo or self.fileno is None or self.ring is None:
            for i in range(count):
                self.pipe_device(device_ptr + i*stride, turbo=turbo, fence=fence)
                self.update()
            return
        self._check_encoder()
        N.check(N.lib().sfx_ring_pipe_frames(self.ring, C.c_void_p(device_ptr), stride, count, self.frame % self.slots, -1 if fence is None else fence, self.fileno))
        self.frame += count

    # finish ---------------------------------------------------------------------------------------------------------

    def finish(self):
        output = None
        self.scene.context.synchronize()
        self.release_buffers()
        self.scene.context.output_top_down(False)
        if self.process is not None:
            self.process.stdin.close()
            if self.process.wait() != 0:                      # the reference only waits (exporting.py:186-187); say what went wrong
                logger.error(f"FFmpeg exited with status {self.process.returncode}:\n" + self.encoder_output())
            output = self.path
        elif self.kind == "path-ffmpeg":
            output = self.path                                # a rank that does not own the sink
        elif self.kind == "path-raw":
            if self.file is not None:
                self.file.close()
            output = self.path
        elif self.kind == "pipe" and self.file is not None:
            self.file.seek(0)
            output = self.file.read()
            self.file.close()
        self.took = (time.monotonic() - self.start)
        self.log_stats(output)
        return output

    def log_stats(self, output) -> None:
        took = self.took or 1e-9
        logger.info(f"Finished rendering ({output if not isinstance(output, bytes) else f'{len(output)} bytes'}) • "
                    f"took {took:.2f}s at {self.frame/took:.2f} fps | {self.scene.runtime/took:.2f}x realtime, {self.frame} frames")
