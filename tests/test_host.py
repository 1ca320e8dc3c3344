"""
Host logic of the package (no GPU): the python mirror of the reference's classes against the golden vectors, the
module-registration contract, and the C-ABI library's exported symbols. CPU only.
"""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_resolution_fit(golden):
    """resolution.py:9-86 incl. the reference's own assertions (:90-116)"""
    from shaderflow_amd.resolution import Resolution
    nan = lambda v: None if np.isnan(v) else v
    for row in golden("resolution")["cases"]:
        ow, oh, nw, nh, mw, mh, ar, scale, w, h = row
        old = (nan(ow), nan(oh)); new = (nan(nw), nan(nh)); mx = (nan(mw), nan(mh))
        old = tuple(int(v) if v is not None else None for v in old)
        new = tuple(int(v) if v is not None else None for v in new)
        mx = tuple(int(v) if v is not None else None for v in mx)
        got = Resolution.fit(old=old, new=new, max=mx if any(mx) else None, ar=nan(ar), scale=scale)
        assert got == (int(w), int(h)), row
    with pytest.raises(ValueError):
        Resolution.fit(old=(1920, None), new=(1280, None))
    with pytest.raises(ValueError):
        Resolution.fit(old=(None, 1080), new=(None, None))


@pytest.mark.parametrize("tag,fps,sr", [("60_44100", 60.0, 44100), ("30_48000", 30.0, 48000), ("24_44100", 24.0, 44100), ("59.94_44100", 60000/1001, 44100)])
def test_clock_and_chunk_schedule(golden, tag, fps, sr):
    from shaderflow_amd.audio.reader import chunk_schedule
    from shaderflow_amd.scheduler import freewheel_clock
    g = golden("clock")
    frames = len(g[f"dt_{tag}"])
    times, dts, rdts = freewheel_clock(fps, frames)
    assert np.array_equal(times, g[f"time_{tag}"]) and np.array_equal(dts, g[f"dt_{tag}"])      # float64 bit-exact
    total = int(sr*(frames/fps)) + 5000
    assert np.array_equal(chunk_schedule(rdts, sr, 2, total), g[f"tell_{tag}"])


def test_reader_stream_matches_schedule(golden):
    from shaderflow_amd.audio.reader import BrokenAudioReader, chunk_schedule
    g = golden("clock")
    pcm = np.zeros((2000, 2), np.float32)
    reader = BrokenAudioReader(samples=pcm, samplerate=44100)
    stream = reader.stream
    lengths = []
    for k in range(6):
        reader.chunk = 0.0 if k == 0 else 1/60
        try:
            lengths.append(next(stream).shape[0])
        except StopIteration:
            lengths.append(-1)
    assert lengths == list(g["len_eof"])
    assert list(chunk_schedule([0.0] + [1/60]*5, 44100, 2, 2000)) == [1, 735, 1470, 2000, 2000, 2000]


def test_wav_round_trip(tmp_path):
    from shaderflow_amd.audio.reader import read_wav, write_wav_f32
    rng = np.random.default_rng(0)
    pcm = rng.uniform(-1, 1, (1000, 2)).astype(np.float32)
    path = write_wav_f32(tmp_path/"a.wav", pcm, 48000)
    back, sr = read_wav(path)
    assert sr == 48000 and np.array_equal(back, pcm)
    with pytest.raises(ValueError):
        (tmp_path/"b.wav").write_bytes(b"not a wav")
        read_wav(tmp_path/"b.wav")


@pytest.mark.parametrize("tag,dtype", [("spec", np.float32), ("volume", None), ("std", None), ("resp", None), ("cosh", None), ("idle", None), ("vardt", None)])
def test_dynamic_number_host_mirror_bit_exact(golden, tag, dtype):
    """shaderflow_amd.dynamics.DynamicNumber against the reference's trajectories (dynamics.py:197-250)"""
    from shaderflow_amd.dynamics import DynamicNumber
    g = golden("dynamics")
    freq, zeta, resp, integ = (float(v) for v in g[f"{tag}_params"])      # python floats: numpy scalars are not "weak"
    targets = g[f"{tag}_targets"]
    if dtype is np.float32:
        system = DynamicNumber(frequency=freq, zeta=zeta, response=resp, dtype=np.float32)
        system.set(np.zeros(targets[0].shape, np.float32))
    else:
        system = DynamicNumber(value=(0.25 if tag == "idle" else 0), frequency=freq, zeta=zeta, response=resp, integrate=bool(integ))
        if tag in ("volume", "std"):
            system.set(system.initial, instant=True)
    for k, dt in enumerate(g[f"{tag}_dts"]):
        system.target = targets[k] if dtype is np.float32 else (np.float32(targets[k]) if tag in ("volume", "std") else float(targets[k]))
        system.next(dt=abs(float(dt)))
        assert np.array_equal(np.asarray(system.value), g[f"{tag}_values"][k]), (tag, k)
        assert np.array_equal(np.asarray(system.integral), g[f"{tag}_integrals"][k]), (tag, k)


def test_filterbank_matrix_and_notes(golden):
    from shaderflow_amd.audio.module import BrokenAudio
    from shaderflow_amd.audio.spectrogram import BrokenSpectrogram, SpectrogramScale
    from shaderflow_amd.piano import PianoNote
    f = golden("filterbank")
    spec = BrokenSpectrogram(audio=BrokenAudio())
    spec.from_notes(start=PianoNote.from_frequency(20), end=PianoNote.from_frequency(14000), piano=True)
    assert spec.spectrogram_bins == 115
    m = spec.spectrogram_matrix()
    assert np.array_equal(m.indptr, f["piano115_indptr"]) and np.array_equal(m.indices, f["piano115_indices"])
    assert np.array_equal(m.data, f["piano115_data"])
    mel = BrokenSpectrogram(audio=BrokenAudio(), scale=SpectrogramScale.MEL)
    mel.spectrogram_bins = 64
    assert np.array_equal(mel.spectrogram_matrix().data, f["mel64_data"])
    assert [PianoNote.frequency_to_index(float(x)) for x in f["note_of_freq_in"]] == list(f["note_of_freq_out"])


def test_broken_audio_window_semantics():
    """get_last_n_samples excludes the newest sample; history starts as zeros (audio/module.py:110-138)"""
    from shaderflow_amd.audio.module import BrokenAudio
    audio = BrokenAudio()
    audio.add_data(np.arange(1, 11, dtype=np.float32)[None, :].repeat(2, 0))
    assert audio.tell == 10
    assert list(audio.get_last_n_samples(4)[0]) == [6, 7, 8, 9]
    assert list(audio.get_last_n_samples(12)[0]) == [0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9]
    assert audio.data.shape == (2, 1323000) and audio.data[0, -1] == 10 and audio.data[0, -11] == 0


def test_module_contract_without_gpu():
    """Registration order, weakref scene, relay, module-without-scene error (module.py:33-53)"""
    from shaderflow_amd.dynamics import ShaderDynamics
    from shaderflow_amd.module import ShaderModule
    from shaderflow_amd.scene import ShaderScene
    from shaderflow_amd.variable import Uniform

    class Scene(ShaderScene):
        def build(self):
            pass

    scene = Scene()
    assert scene.modules == [scene] and scene.name == "Scene"
    a = ShaderDynamics(scene=scene, name="iA", frequency=4, value=0.0)
    b = ShaderDynamics(scene=scene, name="iB", frequency=4, value=np.zeros(3), differentiate=True)
    assert scene.modules[1:] == [a, b] and (a.uuid < b.uuid)
    assert [v.name for v in scene.full_pipeline()][-3:] == ["iA", "iB", "iBDerivative"]
    assert b.type == "vec3" and Uniform("float", "iA") == Uniform("vec2", "iA")
    assert Uniform("vec2", "iResolution").declaration == "uniform vec2 iResolution;"
    with pytest.raises(RuntimeError):
        ShaderModule()
    seen = []
    a.handle = lambda message: seen.append(message)
    scene.relay("ping")
    assert seen == ["ping"]
    assert scene.render_resolution == (1920, 1080)
    scene._ssaa = 1.5
    assert scene.render_resolution == (2880, 1620)


def test_alias_install():
    import shaderflow_amd
    shaderflow_amd.install_alias()
    import shaderflow
    from shaderflow.scene import ShaderScene
    from shaderflow.variable import Uniform
    assert shaderflow.resources.exists() and ShaderScene.__module__ == "shaderflow_amd.scene" and Uniform is not None


def test_alias_covers_every_submodule_with_one_module_object():
    """`shaderflow.<sub>` must be THE module `shaderflow_amd.<sub>` for every submodule (a second import under the alias
    would duplicate classes: isinstance checks between the two copies fail)"""
    import importlib
    import pkgutil

    import shaderflow_amd
    shaderflow_amd.install_alias()
    names = [m.name for m in pkgutil.walk_packages(shaderflow_amd.__path__, "shaderflow_amd.") if not m.name.endswith("libshaderflow_hip")]
    assert {"shaderflow_amd.video", "shaderflow_amd.ffmpeg", "shaderflow_amd.exporting", "shaderflow_amd.piano.module"} <= set(names)
    for real in names:
        alias = "shaderflow" + real[len("shaderflow_amd"):]
        assert importlib.import_module(alias) is importlib.import_module(real), real
    with pytest.raises(ImportError):
        importlib.import_module("shaderflow.no_such_module")


REFERENCE_DEMO = Path("/root/reference/examples/basic/demo.py")


@pytest.mark.skipif(not REFERENCE_DEMO.exists(), reason="the reference checkout is not present on this machine")
def test_the_references_own_demo_imports_unchanged():
    """The drop-in claim for examples/basic/demo.py: the reference's own file, read where it lies and executed unchanged under
    install_alias(), defines its scenes on THIS package's classes (no GPU needed to import and subclass)"""
    import importlib.util

    import shaderflow_amd
    shaderflow_amd.install_alias()
    spec = importlib.util.spec_from_file_location("reference_demo", REFERENCE_DEMO)
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    from shaderflow_amd.scene import ShaderScene
    scenes = [getattr(demo, n) for n in ("Basic", "ShaderToy", "MultiShader", "Multipass", "MotionBlur", "Dynamics", "Video", "Audio",
                                         "Waveform", "MusicBars", "Visualizer", "RayMarch", "Life")]
    assert all(issubclass(scene, ShaderScene) for scene in scenes)
    assert demo.shaders == REFERENCE_DEMO.parent/"shaders"
    # the lazy imports inside the scenes' build() methods resolve too (demo.py:134-137, 149-151, 164-166, 178-182, 196-200)
    from shaderflow.audio import ShaderAudio                          # noqa: F401
    from shaderflow.audio.spectrogram import ShaderSpectrogram        # noqa: F401
    from shaderflow.audio.waveform import ShaderWaveform              # noqa: F401
    from shaderflow.piano import PianoNote                            # noqa: F401
    from shaderflow.video import ShaderVideo
    import shaderflow_amd.video
    assert ShaderVideo is shaderflow_amd.video.ShaderVideo


def test_library_exports_every_declared_symbol():
    """include/shaderflow_hip.h ↔ libshaderflow_hip.so ↔ the ctypes prototype table"""
    from shaderflow_amd import _native as N
    header = (ROOT/"include"/"shaderflow_hip.h").read_text()
    declared = set(re.findall(r"\b(sfx_[a-z0-9_]+)\s*\(", header))
    assert declared == set(N.PROTOTYPES), declared ^ set(N.PROTOTYPES)
    lib = C.CDLL(str(N.LIBRARY))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sfx_fused_supported(2000, 2) == 1 and lib.sfx_fused_supported(1000, 2) == 0     # pure host logic


def test_no_device_fails_loudly():
    """No silent CPU fallback: without a GPU a context cannot be created"""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible")
    from shaderflow_amd import _native as N
    with pytest.raises(N.NativeError, match="no HIP device"):
        N.Context(0)


def test_product_never_imports_the_oracle():
    for path in (ROOT/"shaderflow_amd").rglob("*"):
        if path.suffix in (".py", ".hpp", ".hip", ".h") and path.is_file():
            text = path.read_text()
            assert "oracle" not in text.replace("parity oracle", "").replace("CPU oracle", "").replace("the oracle", ""), path


def test_written_out_freewheel_clock_equals_the_scheduler_task():
    """scheduler.freewheel_clock writes SchedulerTask.next's float64 operations out (no object per frame) and keeps the longest
    sequence per rate: bit-identical to stepping a real SchedulerTask, for prefixes and for rates whose period is not a binary fraction"""
    from shaderflow_amd.scheduler import freewheel_clock, freewheel_clock_by_task
    for fps, frames, speed in ((60.0, 3600, 1.0), (60.0, 100, 1.0), (30, 777, 1.0), (59.94, 2000, 0.5), (24, 100, 2.0), (144.0, 1500, 1.0), (60.0, 4000, 1.0)):
        want = freewheel_clock_by_task(fps, frames, speed)
        got = freewheel_clock(fps, frames, speed)
        assert all(list(g) == list(w) for g, w in zip(got, want)), (fps, frames, speed)


def test_module_skipping_keeps_the_order_in_which_modules_override_each_other():
    """shader.py use_scene_pipeline: a module whose pipeline_token() is unchanged is not walked again — unless a module walked earlier
    in the frame yields one of its names (then the later module has to say its value again: last writer wins, as in a full walk)"""
    from shaderflow_amd.shader import ShaderProgram

    class Module:
        def __init__(self, name, values, token):
            self.name, self.values, self.token, self.walks = name, values, token, 0

        def pipeline_token(self):
            return self.token

        def pipeline(self):
            self.walks += 1
            from shaderflow_amd.variable import Uniform
            return [Uniform("float", key, value) for key, value in self.values.items()]

    class Scene:
        modules: list = []

    first, second, third = Module("a", {"shared": 1.0, "own": 5.0}, None), Module("b", {"shared": 2.0}, "t0"), Module("c", {"other": 3.0}, "t0")
    scene = Scene()
    scene.modules = [first, second, third]
    program = ShaderProgram.__new__(ShaderProgram)
    program.__dict__.update(scene=scene, _module_tokens={}, _module_names={}, _shared_names=frozenset(), sent=[])
    program.use_pipeline = lambda variables: program.sent.extend((v.name, v.value) for v in variables)
    program.use_scene_pipeline()
    assert [m.walks for m in scene.modules] == [1, 1, 1] and program._shared_names == {"shared"}
    program.sent.clear()
    program.use_scene_pipeline()                                      # `first` has no token: walked; it touches "shared", so `second` follows; `third` rests
    assert [m.walks for m in scene.modules] == [2, 2, 1] and program.sent == [("shared", 1.0), ("own", 5.0), ("shared", 2.0)]
    first.token = "fixed"
    program.use_scene_pipeline(); program.sent.clear(); program.use_scene_pipeline()
    assert [m.walks for m in scene.modules] == [3, 3, 1] and program.sent == []          # everything known and unchanged: nothing walked
    third.token = "t1"
    program.use_scene_pipeline()
    assert [m.walks for m in scene.modules] == [3, 3, 2] and program.sent == [("other", 3.0)]


def test_issue_model_prices_the_uncounted_scalar_source_and_packed_forms():
    """bench.py roofline.issue_model (ONE number): class counters x their prices + the uncounted forms at the
    census' price + two cycles per full-rate form with a scalar source and per packed-f32 form (the class counters count a
    v_pk_fma_f32 once, as an fma), over the SIMD cycles the launch had. Hand-computed."""
    import bench
    cs = {"SQ_INSTS_VALU": 1000.0, "SQ_INSTS_VALU_ADD_F32": 200.0, "SQ_INSTS_VALU_MUL_F32": 100.0, "SQ_INSTS_VALU_FMA_F32": 400.0,
          "SQ_INSTS_VALU_INT32": 50.0, "SQ_INSTS_VALU_CVT": 30.0, "SQ_INSTS_VALU_TRANS_F32": 20.0, "GRBM_GUI_ACTIVE": 8.0*4.0, "SQ_THREAD_CYCLES_VALU": 1000.0*64.0}
    census = {"other_cycles_per_instruction": 3.0, "sgpr_source_full_rate_forms_per_valu_instruction": 0.10, "packed_f32_forms_per_valu_instruction": 0.08}
    model = bench.issue_model(cs, 2.0, census)
    had = 4.0*256*4                                                   # GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs
    priced = 200*2 + 100*2 + 400*2 + 50*4 + 30*4 + 20*8
    other = 1000 - (200 + 100 + 400 + 50 + 30 + 20)
    assert model["instructions"]["other"] == other and model["simd_cycles_available"] == had
    assert model["frac_census"] == round((priced + 3.0*other + 2.0*0.10*1000 + 2.0*0.08*1000)/had, 4) == model["frac"]
    assert model["frac_low"] == round((priced + 2*other)/had, 4) and model["frac_other_at_4"] == round((priced + 4*other)/had, 4)
    assert model["effective_clock_GHz"] == 2.0 and model["lane_utilisation"] == 1.0
    assert bench.issue_model(cs, 2.0, None)["frac_census"] is None   # no census for this kernel: the 2-or-4 band stays


def test_isa_census_classes_and_scalar_sources():
    """tools/isa_census.py (the evidence behind issue_model's one number): which class an instruction is priced in, and which
    full-rate forms read a scalar register"""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent/"tools"))
    import isa_census as census
    cases = {"v_fmac_f32": "fma_f32", "v_fmaak_f32": "fma_f32", "v_add_f32": "add_f32", "v_pk_fma_f32": "pk_fma_f32", "v_pk_add_f32": "pk_add_f32",
             "v_rcp_f32": "trans", "v_cvt_pk_u8_f32": "cvt", "v_med3_f32": "minmax_med", "v_cmp_gt_f32": "cmp", "v_cndmask_b32": "cndmask",
             "v_mov_b32": "mov", "v_readfirstlane_b32": "readlane", "v_add_u32": "int_add_logic", "v_lshlrev_b32": "int_shift_mul",
             "ds_read_b128": "lds", "s_load_dwordx4": "smem", "s_waitcnt": "waitcnt", "s_cbranch_scc1": "branch", "s_add_u32": "salu",
             "global_store_dwordx3": "vmem"}
    for mnemonic, want in cases.items():
        assert census.classify(mnemonic, "") == want, (mnemonic, census.classify(mnemonic, ""))
    assert census.classify("v_mov_b32", "v1, v2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") == "dpp_swizzle"
    assert census.uses_sgpr_source("v_mov_b32", "v8, s9") and census.uses_sgpr_source("v_fmac_f32", "v50, s8, v32")
    assert census.uses_sgpr_source("v_pk_fma_f32", "v[38:39], s[8:9], v[8:9], v[38:39] op_sel_hi:[0,1,1]")
    assert not census.uses_sgpr_source("v_fmac_f32", "v50, v52, v13") and not census.uses_sgpr_source("v_readfirstlane_b32", "s22, v1")
    cycles = {name: c for name, _, c in census.CLASSES}
    assert cycles["fma_f32"] == 2 and cycles["pk_fma_f32"] == 4 and cycles["trans"] == 8 and cycles["cvt"] == 4


def test_clock_sequence_chunks_are_sized_by_time():
    """the native clock sequence keeps the host for about a quarter of a second at most — scene.quit, the encoder check and
    Ctrl-C are looked at between chunks — whatever a frame costs; 240 frames only when they are that cheap"""
    from shaderflow_amd.clockloop import ClockLoop
    loop = ClockLoop.__new__(ClockLoop)                            # (chunk_frames reads class constants only)
    assert loop.chunk_frames(None) == 30 and loop.chunk_frames(0.0) == 30          # nothing measured yet: a short first chunk
    assert loop.chunk_frames(0.0005) == 240                                        # 0.5 ms per frame: the cap
    assert loop.chunk_frames(0.004) == 62                                          # 4 ms per frame (a sink that waits): 0.25 s worth
    assert loop.chunk_frames(1.0) == 1                                             # a stalled encoder: frame by frame
