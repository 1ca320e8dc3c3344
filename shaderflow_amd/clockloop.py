"""
ClockLoop: the frame loop of scenes in which nothing but the clock moves (no reference equivalent).

`ShaderScene.next` (scene.py:456-479) updates every module and lets every program walk every module's `pipeline()` — ≈ 150 µs of
python per frame whatever the scene does. Scenes made only of stock modules whose `DynamicNumber`s are at rest — Multipass, MotionBlur,
Life, any layered / temporal feedback scene without python logic between frames — change, from one frame to the next, exactly:
the four clock uniforms (iTime, iTau, iDeltatime, iFrame: scene.py:687-703), `iLayer` per draw, and WHICH device texture sits
behind every sampler of a temporal texture after its `roll()` (texture.py:295-298, 351-381). This loop does that and nothing else
per frame: one `sfx_uniform_set_clock`, one `sfx_sampler_bind_many` per rolled texture matrix, the draws, the resolve, the read-out.
Same launches in the same order with the same uniform values as the frame loop, so the same frames byte for byte
(tests/test_gpu_multipass.py); anything it is not sure about takes the frame loop.
"""
from __future__ import annotations

import ctypes as C
from typing import TYPE_CHECKING

import numpy as np

from shaderflow_amd import _native as N
from shaderflow_amd.camera import CameraMode, ShaderCamera
from shaderflow_amd.dynamics import ShaderDynamics
from shaderflow_amd.module import ShaderModule
from shaderflow_amd.scheduler import freewheel_clock
from shaderflow_amd.shader import ShaderProgram
from shaderflow_amd.texture import ShaderTexture

if TYPE_CHECKING:
    from shaderflow_amd.exporting import ExportingHelper
    from shaderflow_amd.scene import ShaderScene


class ClockLoop:
    @staticmethod
    def applicable(scene: "ShaderScene") -> bool:
        from shaderflow_amd.scene import ShaderScene
        if type(scene).update is not ShaderModule.update:
            return False
        # a scene with a frame step, a message handler or scheduled tasks of its own keeps the ordinary loop (this one runs none of them)
        if type(scene).next is not ShaderScene.next or type(scene).handle is not ShaderScene.handle:
            return False
        if any(task is not scene.vsync for task in scene.scheduler.tasks):
            return False
        if type(scene).pipeline is not ShaderScene.pipeline and not ClockLoop.pipeline_is_static(scene):
            return False
        for module in scene.modules:
            if module is scene or isinstance(module, ShaderScene):
                continue
            if type(module) not in (ShaderCamera, ShaderDynamics, ShaderProgram, ShaderTexture):
                return False
            if isinstance(module, ShaderCamera) and module.mode == CameraMode.Spherical:
                return False                                          # align() runs every update() there and may move rotation.target again (camera.py:220-236)
            if isinstance(module, ShaderDynamics):
                # at rest, and staying there: the early-out of dynamics.py:222-225 — and no integral that keeps running
                value, target = np.asarray(module.value), np.asarray(module.target)
                if module.integrate or value.shape != target.shape or value.dtype.kind not in "fiu":
                    return False
                if value.size and float(np.abs(target - value).max()) >= module.precision:
                    return False
        return True

    CLOCK = ("iTime", "iTau", "iDeltatime", "iFrame")

    @staticmethod
    def pipeline_is_static(scene: "ShaderScene") -> bool:
        """A scene that overrides pipeline() to ADD uniforms (demo.py's Life: `iLifePeriod`): fine as long as what it adds does not move
        with the clock — asked the way the frame loop asks, by calling pipeline(), at two different times"""
        def snapshot():
            out = []
            for variable in scene.pipeline():
                if variable.name in ClockLoop.CLOCK:
                    continue
                value = variable.value
                if isinstance(value, np.ndarray):
                    value = (value.dtype.str, value.shape, value.tobytes())
                elif isinstance(value, (list, tuple)):
                    value = tuple(float(v) if isinstance(v, (int, float, np.number)) else repr(v) for v in value)
                elif not isinstance(value, (int, float, str, bool, type(None))):
                    value = id(value)                                  # a texture: the same object or not
                out.append((variable.type, variable.name, value))
            return out
        saved = (scene.time, scene.dt, scene.rdt)
        try:
            first = snapshot()
            scene.time, scene.dt, scene.rdt = saved[0] + 1.2345, 0.0173, 0.0173
            second = snapshot()
        except Exception:
            return False
        finally:
            scene.time, scene.dt, scene.rdt = saved
        return first == second

    def __init__(self, scene: "ShaderScene"):
        self.scene = scene
        self.programs = [m for m in reversed(scene.modules) if isinstance(m, ShaderProgram)]      # the order scene.next renders them in
        # texture matrices whose samplers move every frame: (names as a C array, the texture) — the names never change
        self.rolling = []
        for module in scene.modules:
            if isinstance(module, ShaderTexture) and module.name and module.temporal > 1:
                names = [module._sampler_name(t, l).encode() for (t, l, _) in module.boxes]
                self.rolling.append((module, (C.c_char_p*len(names))(*names), len(names)))

    def bind_rolled(self, program: ShaderProgram) -> None:
        for texture, names, count in self.rolling:
            handles = (N.Handle*count)(*[box.texture.handle if box.texture is not None else N.Handle() for (_, _, box) in texture.boxes])
            N.check(N.lib().sfx_sampler_bind_many(program.program, names, handles, count))

    # the same loop without python between the frames (csrc/capi.hip sfx_clock_sequence_run) ---------------------------------------------

    CHUNK = 240                                                        # most frames per native call: the encoder and `scene.quit` are looked at in between
    CHUNK_SECONDS = 0.25                                               # … and about how long a call may keep the host: the chunk is sized by the measured frame time

    def chunk_frames(self, measured: "float | None") -> int:
        """Frames of the next native call: CHUNK while nothing is known, then what fits CHUNK_SECONDS at the rate the last call ran at — a 4K
        frame that waits for its ring slot takes 0.5 ms, so 240 of them held scene.quit, the encoder check and Ctrl-C for 120 ms at best and
        for as long as the sink stalls at worst"""
        if not measured or measured <= 0.0:
            return min(self.CHUNK, 30)
        return max(1, min(self.CHUNK, int(self.CHUNK_SECONDS/measured)))

    def native_sequence(self, export: "ExportingHelper", turbo: bool) -> bool:
        """Whether the frames can be rendered, rolled, resolved, read out and piped by ONE native call per chunk: a turbo export without a
        progress relay (a relay wants a python call per frame), every program compiled. SHADERFLOW_CLOCK_SEQUENCE=0 keeps the python loop
        (A/B measurements, and the byte-equality test of the two)."""
        import os
        if os.environ.get("SHADERFLOW_CLOCK_SEQUENCE", "1") == "0" or not turbo or export.relay is not None:
            return False
        return all(program.program is not None for program in self.programs)

    def run_native(self, export: "ExportingHelper", times, dts, rdts, total: int) -> None:
        scene, lib = self.scene, N.lib()
        runtime, fps = scene.runtime, scene.fps
        # texture matrices: every program's own (its draws go to row 0) — the same objects `rolling` lists when they are temporal
        textures = [program.texture for program in self.programs]
        index = {id(texture): m for m, texture in enumerate(textures)}
        passes = (N.SequencePass*len(self.programs))()
        for k, program in enumerate(self.programs):
            if program.texture.final:
                passes[k] = N.SequencePass(N.Handle(), N.PASS_RESOLVE, index[id(scene.shader.texture)], program.texture.fbo.handle, 0, scene.subsample)
            elif program is scene.shader and scene._can_fuse(program):
                passes[k] = N.SequencePass(program.program, N.PASS_FUSED, index[id(program.texture)], scene._final.texture.fbo.handle, int(scene.ssaa), scene.subsample)
            else:
                passes[k] = N.SequencePass(program.program, N.PASS_LAYERS, index[id(program.texture)], N.Handle(), 0, 0)
        keep = []                                                      # ctypes arrays the structures point into

        def matrix_tables():
            tables = (N.SequenceMatrix*len(textures))()
            for m, texture in enumerate(textures):
                boxes = [box for (_, _, box) in texture.boxes]
                handles = (N.Handle*len(boxes))(*[box.texture.handle if box.texture is not None else N.Handle() for box in boxes])
                names = None
                if texture.name and texture.temporal > 1:
                    names = (C.c_char_p*len(boxes))(*[texture._sampler_name(t, l).encode() for (t, l, _) in texture.boxes])
                keep.extend([handles, names])
                tables[m] = N.SequenceMatrix(texture.temporal, texture.layers, handles, names)
            return tables
        planar = None
        if export.planar and export._yuv_slots:
            planar = (C.c_void_p*len(export._yuv_slots))(*export._yuv_slots)
        piping = export.fileno is not None and export.ring is not None
        done = 0
        per_frame = None                                               # seconds per frame of the last native call
        import time as clock
        while done < total and not scene.quit:
            count = min(self.chunk_frames(per_frame), total - done)
            export._check_encoder()
            started = clock.perf_counter()
            ticks = (N.ClockTick*count)()
            for i in range(count):
                time = times[done + i]
                ticks[i] = N.ClockTick(time, (time/runtime) % 1.0, dts[done + i], round(time*fps))
            N.check(lib.sfx_clock_sequence_run(scene.context.handle, passes, len(self.programs), matrix_tables(), len(textures), ticks, count,
                                               export.ring if piping else N.Handle(), export.frame % max(1, export.slots), export.fileno if piping else -1,
                                               planar, 1 if export.yuv_matrix == "bt709" else 0, scene.width, scene.height))
            per_frame = (clock.perf_counter() - started)/count
            for texture in textures:
                texture.roll(count)                                   # the native call rolled its own copy of every matrix it drew into
            export.frame += count
            done += count
            keep.clear()
        if done:
            scene.time, scene.dt, scene.rdt = times[done - 1], dts[done - 1], rdts[done - 1]

    def run(self, export: "ExportingHelper", turbo: bool):
        scene = self.scene
        total = export.total_frames
        times, dts, rdts = freewheel_clock(scene.fps, total, scene.speed)
        lib = N.lib()
        # frame 0's state through the ordinary pipeline walk: every uniform and sampler of every program is on the device
        scene.time, scene.dt, scene.rdt = times[0], dts[0], rdts[0]
        for program in self.programs:
            if program.program is None:
                program.compile()
            if not program.texture.final:
                program.use_scene_pipeline()
        runtime, fps = scene.runtime, scene.fps
        if self.native_sequence(export, turbo):
            try:
                self.run_native(export, times, dts, rdts, total)
            finally:
                for program in self.programs:
                    program._pushed.clear(); program._pushed_plain.clear(); program._module_tokens.clear()
            return export.finish()
        try:
            for k in range(total):
                if scene.quit:                                        # ShaderMessage.Window.Close (scene.py:478-480), as the vsync loop honours it
                    break
                time = times[k]
                scene.time, scene.dt, scene.rdt = time, dts[k], rdts[k]
                scene._fused_this_frame = False
                for program in self.programs:
                    if not program.texture.final:
                        N.check(lib.sfx_uniform_set_clock(program.program, time, (time/runtime) % 1.0, dts[k], round(time*fps)))
                        self.bind_rolled(program)
                    program.render(pipeline=False)
                export.pipe(turbo=turbo)
                export.update()
        finally:
            # what this loop sent behind the programs' backs: their caches of sent values say something older
            for program in self.programs:
                program._pushed.clear(); program._pushed_plain.clear(); program._module_tokens.clear()
        scene.time, scene.dt, scene.rdt = times[-1], dts[-1], rdts[-1]          # the clock of the last frame
        return export.finish()
