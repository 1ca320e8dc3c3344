"""
SURVEY §8 (f3)/(f4) on the GPU: layered and temporal render targets (`<name>{t}x{l}` samplers, iLayer, texelFetch,
float targets), video-as-texture, and the remaining example fragments — HIP against the oracle on identical inputs.
These kernels are the generic one-thread-per-pixel path with the same binary32 operation order as the oracle, so
the bound is bit-exact; scene-level tests replay the reference's render/roll order (shader.py:399-405) on the oracle.
"""
import numpy as np
import pytest

from oracle import binding as O
from shaderflow_amd import synth
from tests.helpers import Gpu, lsb_report

pytestmark = pytest.mark.gpu


@pytest.fixture()
def gpu():
    g = Gpu()
    yield g
    g.close()


def rgba(rng, w, h):
    return rng.integers(0, 256, (h, w, 4), dtype=np.uint8)


# ---- fragments --------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name,kw", [
    ("raymarch", {}), ("raymarch", dict(iCameraPosition=(0.4, 0.2, -1.5), iCameraZoom=0.8)),
    ("raymarch", dict(iCameraProjection=2, iCameraZoom=0.6)),
    ("mandelbrot", {}), ("mandelbrot", dict(iQuality=0.12, iCameraZoom=0.35, iCameraPosition=(-0.3, 0.55, 0.0))),
    ("mandelbrot", dict(iCameraProjection=1)),
    ("tetration", {}), ("tetration", dict(iCameraZoom=2.5, iCameraPosition=(-0.7, 0.1, 0.0))),
])
def test_remaining_example_fragments_bit_exact(gpu, name, kw):
    w, h = 160, 90
    u = O.default_uniforms(w, h, iTime=0.6, iTau=0.06, **kw)
    want = O.render(name, u, {}, w, h, threads=8)
    prog, fallback = gpu.program(name)
    assert not fallback
    gpu.set_uniforms(prog, u)
    got = gpu.render(prog, w, h)
    assert np.array_equal(got, want), lsb_report(got, want)
    assert want[..., :3].std() > 0                                   # not a constant image


def test_multipass_both_layers_bit_exact(gpu):
    w, h = 128, 72
    rng = np.random.default_rng(3)
    background = rng.integers(0, 256, (54, 96, 3), dtype=np.uint8)
    u = O.default_uniforms(w, h)
    prog, fallback = gpu.program("multipass")
    assert not fallback
    gpu.set_uniforms(prog, u)
    assert gpu.bind(prog, "background", gpu.texture(background))
    layer0 = gpu.render(prog, w, h, layer=0)
    u.iLayer = 0
    want0 = O.render("multipass", u, {"background": O.make_texture(background)}, w, h, threads=4)
    assert np.array_equal(layer0, want0), lsb_report(layer0, want0)

    first = gpu.texture(layer0, "linear", False, False)                   # iScreen: linear, clamp (scene.py:192-194)
    assert gpu.bind(prog, "iScreen0x0", first)
    assert not gpu.bind(prog, "iScreen0x1", first)                        # no restated fragment samples layer 1
    layer1 = gpu.render(prog, w, h, layer=1)
    u.iLayer = 1
    want1 = O.render("multipass", u, {"background": O.make_texture(background), 0: O.make_texture(want0, "linear", False, False)}, w, h, threads=8)
    assert np.array_equal(layer1, want1), lsb_report(layer1, want1)
    left, right = layer1[:, :w//2], layer1[:, w//2:]
    assert np.array_equal(left[..., 0], 255 - layer0[:, :w//2, 0]) and np.array_equal(left[..., 1:3], layer0[:, :w//2, 1:3])   # :38-39
    assert not np.array_equal(right, layer0[:, w//2:])                    # blurred half


@pytest.mark.parametrize("temporal", [1, 4, 10, 12])
def test_motionblur_history_bit_exact(gpu, temporal):
    w, h = 96, 54
    rng = np.random.default_rng(temporal)
    history = [rgba(rng, w, h) for _ in range(temporal)]
    u = O.default_uniforms(w, h, iLayer=1)
    u.user[0] = float(temporal)
    textures = {t: O.make_texture(history[t], "linear", False, False) for t in range(temporal)}
    want = O.render("motionblur", u, textures, w, h, threads=4)
    prog, _ = gpu.program("motionblur")
    gpu.set_uniforms(prog, u)
    assert gpu.set_values(prog, "iScreenTemporal", temporal, integer=True)
    for t in range(temporal):
        assert gpu.bind(prog, f"iScreen{t}x0", gpu.texture(history[t], "linear", False, False))
    got = gpu.render(prog, w, h, layer=1)
    assert np.array_equal(got, want), lsb_report(got, want)


def test_motionblur_depth_beyond_the_slots_is_refused(gpu):
    from shaderflow_amd import _native as N
    prog, _ = gpu.program("motionblur")
    gpu.set_uniforms(prog, O.default_uniforms(32, 18))
    gpu.set_values(prog, "iScreenTemporal", 13, integer=True)
    target = gpu.empty(32, 18, 4)
    assert gpu.lib.sfx_render(prog, target, 1) == N.E_UNSUPPORTED


@pytest.mark.parametrize("frame", [0, 6, 7])
def test_life_simulation_float_target_bit_exact(gpu, frame):
    """R32F state, texelFetch with neighbours outside the texture, and the hold branch (iFrame % iLifePeriod != 0)"""
    w, h = 48, 27
    rng = np.random.default_rng(11)
    state = rng.integers(0, 2, (h, w, 1)).astype(np.float32)
    u = O.default_uniforms(w, h, iFrame=frame)
    u.user[0], u.user[1], u.user[2] = w, h, 6
    want = O.render_to("life_simulation", u, {1: O.make_texture(state, "nearest", True, True)}, w, h, 1, np.float32, threads=4)
    prog, _ = gpu.program("life_simulation")
    gpu.set_uniforms(prog, u)
    assert gpu.set_values(prog, "iLifeSize", (w, h)) and gpu.set_values(prog, "iLifePeriod", 6, integer=True)
    assert gpu.bind(prog, "iLife1x0", gpu.texture(state, "nearest", True, True))
    got = gpu.render(prog, w, h, comps=1, dtype=np.float32)
    assert np.array_equal(got, want)
    if frame % 6:
        assert np.array_equal(got, state)                                 # held
    else:                                                                 # Conway's rule with dead cells outside
        padded = np.pad(state[..., 0], 1)
        near = sum(np.roll(np.roll(padded, dy, 0), dx, 1) for dx in (-1, 0, 1) for dy in (-1, 0, 1) if (dx, dy) != (0, 0))[1:-1, 1:-1]
        rule = np.where(state[..., 0] == 1, (near == 2) | (near == 3), near == 3).astype(np.float32)
        assert np.array_equal(got[..., 0], rule)


def test_life_visuals_and_video_bit_exact(gpu):
    w, h = 128, 72
    rng = np.random.default_rng(5)
    states = [rng.integers(0, 2, (27, 48, 1)).astype(np.float32) for _ in range(5)]
    u = O.default_uniforms(w, h, iCameraZoom=0.9)
    want = O.render("life_visuals", u, {t: O.make_texture(states[t], "nearest", True, True) for t in range(5)}, w, h, threads=4)
    prog, _ = gpu.program("life_visuals")
    gpu.set_uniforms(prog, u)
    for t in range(5):
        assert gpu.bind(prog, f"iLife{t}x0", gpu.texture(states[t], "nearest", True, True))
    got = gpu.render(prog, w, h)
    assert np.array_equal(got, want), lsb_report(got, want)

    frame = rng.integers(0, 256, (36, 64, 3), dtype=np.uint8)
    want = O.render("video", u, {0: O.make_texture(frame)}, w, h, threads=4)
    prog, _ = gpu.program("video")
    gpu.set_uniforms(prog, u)
    assert gpu.bind(prog, "iVideo0x0", gpu.texture(frame))
    got = gpu.render(prog, w, h)
    assert np.array_equal(got, want), lsb_report(got, want)


# ---- scenes ------------------------------------------------------------------------------------------------------

def frames_of(raw: bytes, w, h):
    return np.frombuffer(raw, np.uint8).reshape(-1, h, w, 3)


def scene_uniforms(w, h, ssaa, fps, k, times, dts, runtime, **kw):
    return O.default_uniforms(w, h, iTime=times[k], iTau=(times[k]/runtime) % 1.0, iDuration=runtime, iDeltatime=dts[k],
                              iSSAA=float(ssaa), iFramerate=fps, iFrame=round(times[k]*fps), **kw)


def test_multipass_scene_matches_oracle():
    from examples.scenes import Multipass, make
    w, h, fps, seconds, ssaa = 96, 54, 30.0, 0.1, 2
    background = synth.background_image(120, 68, seed=2)
    raw = make(Multipass, background=background).main(width=w, height=h, fps=fps, ssaa=ssaa, time=seconds, output=bytes)
    got = frames_of(raw, w, h)
    assert got.shape[0] == 3
    times, dts, _ = O.clock(fps, 3)
    bg = O.make_texture(np.flipud(background))
    for k in range(3):
        u = scene_uniforms(w, h, ssaa, fps, k, times, dts, seconds, iLayer=0)
        layer0 = O.render("multipass", u, {"background": bg}, w*ssaa, h*ssaa, threads=8)
        u.iLayer = 1
        layer1 = O.render("multipass", u, {"background": bg, 0: O.make_texture(layer0, "linear", False, False)}, w*ssaa, h*ssaa, threads=8)
        want = O.resolve(layer1, w, h, 2, threads=8)
        assert np.array_equal(got[k], want), (k, lsb_report(got[k], want))


def test_motionblur_scene_rolls_like_the_reference():
    """temporal = 10: the matrix rotates after every render (shader.py:405), history fills frame by frame, and iFinal reads
    row 0 AFTER the roll — the frame rendered temporal-1 frames ago (texture.py:253-256, 355-356), black until then."""
    from examples.scenes import MotionBlur, make
    w, h, fps, frames, T = 64, 36, 30.0, 12, 10
    seconds = frames/fps
    background = synth.background_image(80, 45, seed=4)
    scene = make(MotionBlur, background=background)
    raw = scene.main(width=w, height=h, fps=fps, ssaa=1, time=seconds, output=bytes)
    got = frames_of(raw, w, h)
    assert got.shape[0] == frames
    times, dts, _ = O.clock(fps, frames)
    bg = O.make_texture(np.flipud(background))
    zeros = np.zeros((h, w, 4), np.uint8)
    rows = [[zeros, zeros] for _ in range(T)]                            # rows[t] = [layer 0, layer 1], t frames back
    for k in range(frames):
        u = scene_uniforms(w, h, 1, fps, k, times, dts, seconds, iLayer=0)
        u.user[0] = float(T)
        layer0 = O.render("motionblur", u, {"background": bg}, w, h, threads=4)
        history = [layer0] + [rows[t][0] for t in range(1, T)]           # iScreen0x0 is the layer just rendered
        u.iLayer = 1
        layer1 = O.render("motionblur", u, {t: O.make_texture(history[t], "linear", False, False) for t in range(T)}, w, h, threads=4)
        rows[0] = [layer0, layer1]
        rows = [rows[-1]] + rows[:-1]                                    # deque.rotate(1)
        want = O.resolve(rows[0][1], w, h, 2, threads=4)                 # iScreen = iScreen0x1 after the roll
        assert np.array_equal(got[k], want), (k, lsb_report(got[k], want))
    assert not got[:T - 1].any() and got[T - 1].any()


def test_life_scene_matches_oracle():
    from examples.scenes import Life
    w, h, fps, frames, T, period = 96, 54, 30.0, 9, 10, 3
    seconds = frames/fps
    np.random.seed(7)
    scene = type("Life", (Life,), {"life_period": period})()
    raw = scene.main(width=w, height=h, fps=fps, ssaa=1, time=seconds, output=bytes)
    got = frames_of(raw, w, h)
    np.random.seed(7)
    initial = np.random.randint(0, 2, (192, 108), dtype=bool).astype(np.float32).reshape(108, 192, 1)      # the bytes, as rows of 192
    times, dts, _ = O.clock(fps, frames)
    zeros = np.zeros((108, 192, 1), np.float32)
    rows = [zeros.copy() for _ in range(T)]
    rows[1] = initial                                                    # texture.write(..., temporal=1), demo.py:233
    for k in range(frames):
        u = scene_uniforms(192, 108, 1, fps, k, times, dts, seconds)     # same scene uniforms; the target is the life texture
        u.iResolution[0], u.iResolution[1] = w, h
        u.user[0], u.user[1], u.user[2] = 192, 108, period
        rows[0] = O.render_to("life_simulation", u, {1: O.make_texture(rows[1], "nearest", True, True)}, 192, 108, 1, np.float32, threads=4)
        rows = [rows[-1]] + rows[:-1]
        uv = scene_uniforms(w, h, 1, fps, k, times, dts, seconds)
        screen = O.render("life_visuals", uv, {t: O.make_texture(rows[t], "nearest", True, True) for t in range(5)}, w, h, threads=4)
        want = O.resolve(screen, w, h, 2, threads=4)
        assert np.array_equal(got[k], want), (k, lsb_report(got[k], want))
    assert got.std() > 1


def test_video_scene_uploads_frames_on_time():
    from examples.scenes import Video
    w, h, fps = 64, 36, 60.0
    rng = np.random.default_rng(1)
    clip = rng.integers(0, 256, (3, 18, 32, 3), dtype=np.uint8)
    scene = type("Video", (Video,), {"clip": (clip, 30.0)})()
    raw = scene.main(width=w, height=h, fps=fps, ssaa=1, time=6/fps, output=bytes)
    got = frames_of(raw, w, h)
    times, dts, _ = O.clock(fps, 6)
    shown, read = None, 0
    for k in range(6):
        if read < len(clip) and times[k] > read/30.0:                    # video.py:60
            shown, read = clip[read], read + 1
        u = scene_uniforms(w, h, 1, fps, k, times, dts, 6/fps)
        frame = np.flipud(shown) if shown is not None else np.zeros((18, 32, 3), np.uint8)
        want = O.resolve(O.render("video", u, {0: O.make_texture(frame)}, w, h), w, h, 2)
        assert np.array_equal(got[k], want), (k, lsb_report(got[k], want))
    assert read == 3


@pytest.mark.parametrize("name", ["MotionBlur", "Video", "Life"])
def test_sharded_frame_loop_reproduces_the_single_process_frames(name):
    """SURVEY §8e for frame-loop scenes: two ranks' shares (run here one after the other, `shard=(rank, 2)`) put together
    are the single-process export — temporal history through warm-up frames (MotionBlur: 2*(10-1)), host logic on every
    frame (Video uploads), unbounded feedback by rendering everything before (Life: shard_warmup=None)."""
    import examples.scenes as scenes
    w, h, fps, frames, batch = 64, 36, 30.0, 40, 30
    rng = np.random.default_rng(1)
    clip = rng.integers(0, 256, (24, 18, 32, 3), dtype=np.uint8)

    def build():
        base = getattr(scenes, name)
        attrs = {"clip": (clip, 20.0)} if name == "Video" else ({"shard_warmup": None, "life_period": 2} if name == "Life" else {})
        np.random.seed(3)
        return type(name, (base,), attrs)()

    kw = dict(width=w, height=h, fps=fps, ssaa=1, time=frames/fps, output=bytes)
    whole = frames_of(build().main(**kw), w, h)
    assert whole.shape[0] == frames
    parts = [frames_of(build().main(shard=(rank, 2), **kw), w, h) for rank in range(2)]
    assert parts[0].shape[0] == 30 and parts[1].shape[0] == 10              # batches of 30 frames: [0, 30) and [30, 40)
    assert np.array_equal(parts[0], whole[:30]), lsb_report(parts[0], whole[:30])
    assert np.array_equal(parts[1], whole[30:]), lsb_report(parts[1], whole[30:])
    assert whole[30:].std() > 0


@pytest.mark.parametrize("name", ["Multipass", "MotionBlur", "Life"])
def test_clock_loop_gives_the_frame_loops_bytes(name, monkeypatch):
    """clockloop.ClockLoop — scenes in which only the clock moves between frames (layers, temporal history, a second program, a scene
    pipeline() that adds a constant uniform) — against scene.next's loop: the same launches with the same uniforms, so the same bytes;
    and it is really the loop that ran"""
    import examples.scenes as scenes
    from shaderflow_amd.clockloop import ClockLoop
    kw = dict(width=96, height=54, fps=30.0, ssaa=1, time=14/30, output=bytes)
    if name == "Multipass":
        kw["ssaa"] = 2

    def build(clock_loop: bool):
        np.random.seed(11)                                            # Life seeds its first generation from numpy's global state
        scene = scenes.make(getattr(scenes, name), background=synth.background_image(120, 68, seed=4)) if name != "Life" else scenes.Life()
        scene.clock_loop = clock_loop
        return scene
    runs = []
    original = ClockLoop.run
    monkeypatch.setattr(ClockLoop, "run", lambda self, export, turbo: (runs.append(name), original(self, export, turbo))[1])
    natives = []
    native = ClockLoop.run_native
    monkeypatch.setattr(ClockLoop, "run_native", lambda self, *args: (natives.append(name), native(self, *args))[1])
    lean = build(True).main(**kw)                                     # the native sequence: one C call per chunk of frames
    assert runs == [name] and natives == [name]
    monkeypatch.setenv("SHADERFLOW_CLOCK_SEQUENCE", "0")
    python = build(True).main(**kw)                                   # the same loop with python between the frames
    assert runs == [name, name] and natives == [name]
    loop = build(False).main(**kw)
    assert runs == [name, name] and len(lean) == len(loop) == 14*96*54*3
    assert lean == loop, f"{np.count_nonzero(np.frombuffer(lean, np.uint8) != np.frombuffer(loop, np.uint8))} bytes differ"
    assert python == loop
    # … and as yuv420p (the conversion and the read-out of every frame inside the native call)
    monkeypatch.delenv("SHADERFLOW_CLOCK_SEQUENCE")
    planar = build(True).main(pixel_format="yuv420p", **kw)
    assert natives == [name, name] and planar == build(False).main(pixel_format="yuv420p", **kw) and len(planar) == 14*96*54*3//2
    # a chunk boundary in the middle of the temporal history (MotionBlur keeps ten frames): three frames per native call
    monkeypatch.setattr(ClockLoop, "CHUNK", 3)
    assert build(True).main(**kw) == loop
    # a scene with python logic between frames never qualifies, nor one whose DynamicNumbers are still moving
    scripted = scenes.make(scenes.Dynamics, background=synth.background_image(120, 68, seed=4))
    scripted.initialize()
    assert not ClockLoop.applicable(scripted)
    moving = build(True)
    moving.initialize()
    moving.camera.zoom.target = 2.0
    assert not ClockLoop.applicable(moving)
