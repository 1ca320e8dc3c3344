"""
ctypes binding of libshaderflow_hip.so (include/shaderflow_hip.h) — the only way the host package reaches
the GPU. There is NO CPU fallback: if the library is missing, or no device is visible when a context is
created, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PACKAGE = Path(__file__).resolve().parent
LIBRARY = PACKAGE/"libshaderflow_hip.so"

OK = 0
U8, F32, U16, F16 = 0, 1, 2, 3
NEAREST, LINEAR = 0, 1
T_FLOAT, T_INT, T_BOOL, T_VEC2, T_VEC3, T_VEC4, T_MAT2, T_MAT3, T_MAT4 = range(9)
TAPE_SPECTROGRAM, TAPE_WAVEFORM, TAPE_UNIFORMS, TAPE_TARGETS, TAPE_LOUDNESS, TAPE_SCROLL = range(6)
E_UNSUPPORTED = -4

Handle = C.c_uint64


def source_fingerprint() -> str:
    """sha256 over the kernel sources the library is built from (csrc/): ties profiles under profiles/ to the code they measured"""
    import hashlib
    digest = hashlib.sha256()
    for path in sorted((PACKAGE/"csrc").iterdir()):
        if path.suffix in (".hip", ".hpp", ".inc") or path.name == "Makefile":
            digest.update(path.name.encode())
            digest.update(path.read_bytes())
    return digest.hexdigest()[:16]


class CtxInfo(C.Structure):
    _fields_ = [("device_name", C.c_char*128), ("gcn_arch", C.c_char*64), ("device_id", C.c_int32),
                ("compute_units", C.c_int32), ("max_texture_dim", C.c_int32), ("wavefront_size", C.c_int32),
                ("total_memory", C.c_int64), ("lds_per_cu", C.c_int64)]


class DynCoeffF32(C.Structure):
    _fields_ = [("dt", C.c_float), ("k1", C.c_float), ("k2", C.c_float), ("k3", C.c_float)]


class DynCoeffF64(C.Structure):
    _fields_ = [("dt", C.c_double), ("k1", C.c_double), ("k2", C.c_double), ("k3", C.c_double)]


class FrameClock(C.Structure):
    _fields_ = [("iTime", C.c_float), ("iTau", C.c_float), ("iSpectrogramOffset", C.c_float), ("iFrame", C.c_int32)]


class SequencePass(C.Structure):
    _fields_ = [("program", Handle), ("kind", C.c_int), ("matrix", C.c_int), ("target", Handle), ("ssaa", C.c_int), ("subsample", C.c_int)]


class SequenceMatrix(C.Structure):
    _fields_ = [("temporal", C.c_int), ("layers", C.c_int), ("textures", C.POINTER(Handle)), ("names", C.POINTER(C.c_char_p))]


class ClockTick(C.Structure):
    _fields_ = [("time", C.c_float), ("tau", C.c_float), ("deltatime", C.c_float), ("frame", C.c_int32)]


PASS_LAYERS, PASS_FUSED, PASS_RESOLVE = 0, 1, 2


class Binding(C.Structure):
    """sfx_binding (include/shaderflow_hip.h): a uniform or sampler name of a loaded program"""
    _fields_ = [("name", C.c_char_p), ("sampler", C.c_int), ("slot", C.c_int), ("count", C.c_int), ("integer", C.c_int)]


class TapeDesc(C.Structure):
    _fields_ = [("points", C.c_int32), ("chunk_size", C.c_int32), ("reducer", C.c_int32), ("volume_window", C.c_int32),
                ("use_mfma", C.c_int32), ("volume_integrate", C.c_int32), ("std_integrate", C.c_int32),
                ("length_samples", C.c_int32), ("precision", C.c_double)]


# name → (restype, argtypes); every symbol declared in include/shaderflow_hip.h
P = C.POINTER
PROTOTYPES: dict[str, tuple] = {
    "sfx_last_error": (C.c_char_p, []),
    "sfx_version": (C.c_char_p, []),
    "sfx_abi_layout": (C.c_uint64, []),
    "sfx_last_kernel": (C.c_char_p, []),
    "sfx_ctx_create": (C.c_int, [C.c_int, C.c_void_p, P(Handle)]),
    "sfx_ctx_info": (C.c_int, [Handle, P(CtxInfo)]),
    "sfx_ctx_synchronize": (C.c_int, [Handle]),
    "sfx_ctx_output_top_down": (C.c_int, [Handle, C.c_int]),
    "sfx_ctx_filter_model": (C.c_int, [Handle, C.c_int]),
    "sfx_ctx_copy_streams": (C.c_int, [Handle, P(C.c_int), P(C.c_int)]),
    "sfx_ctx_tile_misses": (C.c_int, [Handle, P(C.c_ulonglong)]),
    "sfx_ctx_destroy": (C.c_int, [Handle]),
    "sfx_event_record": (C.c_int, [Handle, C.c_int]),
    "sfx_event_elapsed_ms": (C.c_int, [Handle, C.c_int, C.c_int, P(C.c_float)]),
    "sfx_texture_create": (C.c_int, [Handle, C.c_int, C.c_int, C.c_int, C.c_int, P(Handle)]),
    "sfx_texture_params": (C.c_int, [Handle, C.c_int, C.c_int, C.c_int]),
    "sfx_texture_write": (C.c_int, [Handle, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sfx_texture_read": (C.c_int, [Handle, C.c_void_p, C.c_size_t]),
    "sfx_texture_build_mipmaps": (C.c_int, [Handle]),
    "sfx_texture_read_level": (C.c_int, [Handle, C.c_int, C.c_void_p, C.c_size_t]),
    "sfx_texture_device_ptr": (C.c_int, [Handle, P(C.c_void_p), P(C.c_size_t)]),
    "sfx_texture_destroy": (C.c_int, [Handle]),
    "sfx_program_lookup": (C.c_int, [Handle, C.c_char_p, P(Handle), P(C.c_int)]),
    "sfx_program_load": (C.c_int, [Handle, C.c_void_p, C.c_size_t, P(Binding), C.c_int, P(Handle)]),
    "sfx_program_name": (C.c_char_p, [Handle]),
    "sfx_program_fusable": (C.c_int, [Handle, C.c_int]),
    "sfx_uniform_set": (C.c_int, [Handle, C.c_char_p, C.c_int, C.c_void_p, P(C.c_int)]),
    "sfx_sampler_bind": (C.c_int, [Handle, C.c_char_p, Handle, P(C.c_int)]),
    "sfx_sampler_bind_many": (C.c_int, [Handle, P(C.c_char_p), P(Handle), C.c_int]),
    "sfx_uniform_set_clock": (C.c_int, [Handle, C.c_float, C.c_float, C.c_float, C.c_int]),
    "sfx_program_destroy": (C.c_int, [Handle]),
    "sfx_render": (C.c_int, [Handle, Handle, C.c_int]),
    "sfx_resolve": (C.c_int, [Handle, Handle, Handle, C.c_int]),
    "sfx_render_resolve": (C.c_int, [Handle, Handle, C.c_int, C.c_int]),
    "sfx_fused_supported": (C.c_int, [C.c_int, C.c_int]),
    "sfx_ring_create": (C.c_int, [Handle, C.c_size_t, C.c_int, P(Handle)]),
    "sfx_ring_read_async": (C.c_int, [Handle, Handle, C.c_int]),
    "sfx_ring_read_device_async": (C.c_int, [Handle, C.c_void_p, C.c_int]),
    "sfx_ring_fence": (C.c_int, [Handle, C.c_int]),
    "sfx_ring_read_fenced_async": (C.c_int, [Handle, C.c_void_p, C.c_int, C.c_int]),
    "sfx_ring_pipe_frames": (C.c_int, [Handle, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sfx_ring_stream_wait": (C.c_int, [Handle, C.c_int]),
    "sfx_ring_sync": (C.c_int, [Handle, C.c_int, P(C.c_void_p)]),
    "sfx_ring_pipe": (C.c_int, [Handle, C.c_int, C.c_int]),
    "sfx_ring_pipe_sync": (C.c_int, [Handle, C.c_int]),
    "sfx_ring_destroy": (C.c_int, [Handle]),
    "sfx_rgb_to_yuv420": (C.c_int, [Handle, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sfx_peer_export": (C.c_int, [Handle, C.c_void_p, C.c_void_p]),
    "sfx_peer_open": (C.c_int, [Handle, C.c_void_p, P(C.c_void_p)]),
    "sfx_peer_close": (C.c_int, [Handle, C.c_void_p]),
    "sfx_peer_copy": (C.c_int, [Handle, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "sfx_peer_fence": (C.c_int, [Handle, C.c_int]),
    "sfx_peer_flush": (C.c_int, [Handle]),
    "sfx_peer_route": (C.c_int, [Handle, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "sfx_shm_create": (C.c_int, [Handle, C.c_char_p, C.c_int, C.c_int, C.c_size_t, C.c_int, P(Handle)]),
    "sfx_shm_push": (C.c_int, [Handle, C.c_void_p]),
    "sfx_shm_flush": (C.c_int, [Handle]),
    "sfx_shm_wait": (C.c_int, [Handle, C.c_int64]),
    "sfx_shm_drain": (C.c_int, [Handle, C.c_int, P(C.c_int32), P(C.c_int32), C.c_int]),
    "sfx_shm_drain_wait": (C.c_int, [Handle]),
    "sfx_shm_abort": (C.c_int, [Handle]),
    "sfx_shm_unlink": (C.c_int, [Handle]),
    "sfx_shm_destroy": (C.c_int, [Handle]),
    "sfx_flac_info": (C.c_int, [C.c_void_p, C.c_size_t, P(C.c_int64), P(C.c_int), P(C.c_int), P(C.c_int)]),
    "sfx_flac_decode": (C.c_int, [C.c_void_p, C.c_size_t, P(C.c_float), C.c_int64, P(C.c_int64)]),
    "sfx_audio_upload": (C.c_int, [Handle, P(C.c_float), C.c_int64, C.c_int, C.c_int, P(Handle)]),
    "sfx_audio_destroy": (C.c_int, [Handle]),
    "sfx_stft_plan": (C.c_int, [Handle, C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_int32), P(C.c_int32), P(C.c_float), P(Handle)]),
    "sfx_stft_plan_resampled": (C.c_int, [Handle, C.c_int, C.c_int, P(C.c_int32), P(C.c_int32), P(C.c_double), C.c_int, C.c_int, C.c_int,
                                          P(C.c_int32), P(C.c_int32), P(C.c_float), P(Handle)]),
    "sfx_stft_plan_magnitude": (C.c_int, [Handle, C.c_int]),
    "sfx_stft_plan_window": (C.c_int, [Handle, P(C.c_double), C.c_int]),
    "sfx_stft_plan_destroy": (C.c_int, [Handle]),
    "sfx_stft_power": (C.c_int, [Handle, Handle, P(C.c_int64), C.c_int, P(C.c_float)]),
    "sfx_stft_spectrum": (C.c_int, [Handle, Handle, P(C.c_int64), C.c_int, P(C.c_double)]),
    "sfx_filterbank_apply": (C.c_int, [Handle, P(C.c_float), C.c_int, C.c_int, P(C.c_float)]),
    "sfx_spectrogram_targets": (C.c_int, [Handle, Handle, P(C.c_int64), C.c_int, C.c_int, P(C.c_float)]),
    "sfx_waveform_rows": (C.c_int, [Handle, P(C.c_int64), C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_float)]),
    "sfx_volume_std": (C.c_int, [Handle, P(C.c_int64), C.c_int, C.c_int, P(C.c_float)]),
    "sfx_dynamics_scan": (C.c_int, [Handle, C.c_int, C.c_int, P(C.c_float), P(DynCoeffF32), C.c_float, P(C.c_float), P(C.c_float)]),
    "sfx_dynamics_scan_f64": (C.c_int, [Handle, C.c_int, C.c_int, P(C.c_double), P(DynCoeffF64), C.c_double, C.c_int, P(C.c_double), P(C.c_double)]),
    "sfx_tape_create": (C.c_int, [Handle, Handle, P(TapeDesc), C.c_int, P(Handle)]),
    "sfx_clock_tape_create": (C.c_int, [Handle, C.c_int, P(Handle)]),
    "sfx_tape_reset": (C.c_int, [Handle]),
    "sfx_clock_sequence_run": (C.c_int, [Handle, P(SequencePass), C.c_int, P(SequenceMatrix), C.c_int, P(ClockTick), C.c_int, Handle, C.c_int, C.c_int,
                                         P(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    "sfx_tape_build": (C.c_int, [Handle, C.c_int, P(C.c_int64), P(FrameClock), P(DynCoeffF32), P(DynCoeffF64), P(DynCoeffF64)]),
    "sfx_tape_read": (C.c_int, [Handle, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "sfx_tape_destroy": (C.c_int, [Handle]),
    "sfx_render_tape": (C.c_int, [Handle, Handle, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sfx_device_alloc": (C.c_int, [Handle, C.c_size_t, P(C.c_void_p)]),
    "sfx_device_free": (C.c_int, [Handle, C.c_void_p]),
    "sfx_device_copy": (C.c_int, [Handle, C.c_void_p, C.c_void_p, C.c_size_t]),
    "sfx_device_read": (C.c_int, [Handle, C.c_void_p, C.c_void_p, C.c_size_t]),
}

_lib: C.CDLL | None = None


class NativeError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libshaderflow_hip: {message} (code {code})")
        self.code = code


def lib() -> C.CDLL:
    """Loads the HIP library; raises ImportError when it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        path = Path(os.environ.get("SHADERFLOW_HIP_LIBRARY", LIBRARY))
        if not path.exists():
            raise ImportError(
                f"{path} is missing: build it with `make -C shaderflow_amd/csrc` "
                f"(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
            )
        L = C.CDLL(str(path), mode=C.RTLD_GLOBAL)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(L, name)             # AttributeError if the header and the library disagree
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def check(code: int) -> None:
    if code != OK:
        raise NativeError(code, lib().sfx_last_error().decode("utf-8", "replace"))


def as_ptr(array: np.ndarray, ctype):
    return array.ctypes.data_as(C.POINTER(ctype))


NUMPY_DTYPES = {np.dtype(np.uint8): U8, np.dtype(np.float32): F32, np.dtype(np.uint16): U16, np.dtype(np.float16): F16}


class Context:
    """One HIP device + one stream (the GL context of scene.py:145-157)"""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.handle = Handle()
        check(lib().sfx_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self.handle)))
        self.device = device

    def info(self) -> CtxInfo:
        info = CtxInfo()
        check(lib().sfx_ctx_info(self.handle, C.byref(info)))
        return info

    def synchronize(self) -> None:
        check(lib().sfx_ctx_synchronize(self.handle))

    def copy_streams(self) -> tuple[int, int]:
        """(streams looked at, of which in series with the render stream) when the context's two copy streams were chosen"""
        candidates, colliding = C.c_int(), C.c_int()
        check(lib().sfx_ctx_copy_streams(self.handle, C.byref(candidates), C.byref(colliding)))
        return candidates.value, colliding.value

    def tile_misses(self) -> int:
        """Blocks of the LDS-tiled visualizer kernels that ran the generic taps since the previous call (the first call starts the count)"""
        blocks = C.c_ulonglong()
        check(lib().sfx_ctx_tile_misses(self.handle, C.byref(blocks)))
        return blocks.value

    def filter_model(self, model: str | int) -> None:
        """How LINEAR unorm8 textures of this context are filtered (sfx_ctx_filter_model): "spec" — float weights, the default — or
        "llvmpipe" / "fixed8": the 8-bit fixed-point filter of the software rasteriser the reference's CPU path runs on"""
        code = {"spec": 0, "float": 0, "llvmpipe": 1, "fixed8": 1}.get(model.strip().lower()) if isinstance(model, str) else int(model)
        if code is None:
            raise ValueError(f"filter model {model!r}: 'spec' or 'llvmpipe'")
        check(lib().sfx_ctx_filter_model(self.handle, code))

    def output_top_down(self, enabled: bool) -> None:
        check(lib().sfx_ctx_output_top_down(self.handle, 1 if enabled else 0))

    def event_record(self, slot: int) -> None:
        check(lib().sfx_event_record(self.handle, slot))

    def event_elapsed_ms(self, start: int, stop: int) -> float:
        ms = C.c_float()
        check(lib().sfx_event_elapsed_ms(self.handle, start, stop, C.byref(ms)))
        return ms.value

    def alloc(self, nbytes: int) -> int:
        ptr = C.c_void_p()
        check(lib().sfx_device_alloc(self.handle, nbytes, C.byref(ptr)))
        return ptr.value

    def free(self, ptr: int) -> None:
        check(lib().sfx_device_free(self.handle, C.c_void_p(ptr)))

    def rgb_to_yuv420(self, rgb: int, yuv: int, width: int, height: int, frames: int = 1, matrix: int = 0) -> None:
        """`frames` consecutive RGB8 frames on the device → planar yuv420p on the device (sfx_rgb_to_yuv420), on the context's stream"""
        check(lib().sfx_rgb_to_yuv420(self.handle, C.c_void_p(rgb), C.c_void_p(yuv), width, height, frames, matrix))

    def copy(self, dst: int, src: int, nbytes: int) -> None:
        check(lib().sfx_device_copy(self.handle, C.c_void_p(dst), C.c_void_p(src), nbytes))

    # peer windows (sharded export, "device-sdma")
    def peer_export(self, ptr: int) -> bytes:
        handle = C.create_string_buffer(64)
        check(lib().sfx_peer_export(self.handle, C.c_void_p(ptr), handle))
        return handle.raw

    def peer_route(self) -> dict:
        """How this context's peer copies travel: {"route": "sdma-engines" | "hip-streams" | None, "engines": [ids], "copies", "bytes"}"""
        via, ids, copies, nbytes = C.c_int(), (C.c_int*2)(), C.c_ulonglong(), C.c_ulonglong()
        check(lib().sfx_peer_route(self.handle, C.byref(via), ids, C.byref(copies), C.byref(nbytes)))
        route = {1: "sdma-engines", 0: "hip-streams"}.get(via.value)
        return {"route": route, "engines": [int(i) for i in ids] if via.value == 1 else [], "copies": int(copies.value), "bytes": int(nbytes.value)}

    def peer_open(self, handle: bytes) -> int:
        ptr = C.c_void_p()
        check(lib().sfx_peer_open(self.handle, C.create_string_buffer(handle, 64), C.byref(ptr)))
        return ptr.value

    def peer_close(self, ptr: int) -> None:
        check(lib().sfx_peer_close(self.handle, C.c_void_p(ptr)))

    def peer_copy(self, remote: int, local: int, nbytes: int, lane: int = 0) -> None:
        check(lib().sfx_peer_copy(self.handle, C.c_void_p(remote), C.c_void_p(local), nbytes, lane))

    def peer_fence(self, lane: int) -> None:
        check(lib().sfx_peer_fence(self.handle, lane))

    def peer_flush(self) -> None:
        check(lib().sfx_peer_flush(self.handle))

    def read(self, ptr: int, nbytes: int) -> np.ndarray:
        out = np.empty(nbytes, np.uint8)
        check(lib().sfx_device_read(self.handle, C.c_void_p(ptr), out.ctypes.data, nbytes))
        return out

    def destroy(self) -> None:
        if self.handle.value:
            lib().sfx_ctx_destroy(self.handle)
            self.handle = Handle()


_default_context: Context | None = None


def default_context() -> Context:
    """Process-wide context on LOCAL_RANK's device (one process per GPU)"""
    global _default_context
    if _default_context is None:
        _default_context = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_context
