"""
The optional half of the encoder hand-off (exporting.py:94-134 leaves rgb24 → yuv420p to ffmpeg's swscale on the CPU):
planar 4:2:0 made on the device. The arithmetic is the product's own definition (capi_readout.hip k_rgb_to_yuv420: BT.601 / BT.709 limited
range, 8-bit integer coefficients, chroma from the rounded 2x2 mean), restated in the oracle; known answers pin the coefficients.
"""
import numpy as np
import pytest

from oracle import binding as O
from shaderflow_amd import _native as N
from shaderflow_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("matrix", [0, 1])
@pytest.mark.parametrize("size", [(2, 2), (64, 36), (3840, 2160), (510, 258)])
def test_conversion_kernel_equals_the_restatement(size, matrix):
    w, h = size
    rng = np.random.default_rng(w*7 + h + matrix)
    frames = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)
    if size == (2, 2):                                                  # known answers: white, black, saturated red
        frames[0], frames[1], frames[2] = (255, 255, 255), (0, 0, 0), (255, 0, 0)
    import torch
    context = N.default_context()
    rgb = torch.from_numpy(frames).cuda()
    yuv = torch.zeros(frames.nbytes//2, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    context.rgb_to_yuv420(rgb.data_ptr(), yuv.data_ptr(), w, h, frames=3, matrix=matrix)
    context.synchronize()
    got = yuv.cpu().numpy().reshape(3, -1)
    for k in range(3):
        assert np.array_equal(got[k], O.rgb_to_yuv420(frames[k], matrix)), (size, matrix, k)
    if size == (2, 2):
        assert got[0].tolist() == [235]*4 + [128, 128] and got[1].tolist() == [16]*4 + [128, 128]
        assert got[2].tolist() == ([82]*4 + [90, 240] if matrix == 0 else [63]*4 + [102, 240])
    # odd extents are refused, not rounded
    assert N.lib().sfx_rgb_to_yuv420(context.handle, 1, 1, 3, 2, 1, 0) != 0


@pytest.mark.parametrize("batch", [True, False])
def test_export_in_yuv420p_is_the_rgb_export_converted(batch):
    """scene.main(pixel_format="yuv420p") — through the frame tape (a whole batch converted behind its render) and through the frame
    loop (the final texture converted per frame) — delivers exactly the conversion of the frames the rgb24 export delivers, 1.5 bytes
    per pixel; and the encoder's command line asks for yuv420p rawvideo"""
    from examples.scenes import Visualizer, make
    w, h, frames = 128, 72, 7
    pcm, background = synth.sweep_clip(0.5, 44100), synth.background_image(160, 90, seed=3)
    kw = dict(width=w, height=h, fps=60.0, ssaa=2, time=frames/60.0, output=bytes, batch=batch)
    rgb = np.frombuffer(make(Visualizer, audio=(pcm, 44100), background=background).main(**kw), np.uint8).reshape(frames, h, w, 3)
    scene = make(Visualizer, audio=(pcm, 44100), background=background)
    planar = np.frombuffer(scene.main(pixel_format="yuv420p", **kw), np.uint8)
    assert planar.size == frames*w*h*3//2
    planar = planar.reshape(frames, -1)
    for k in range(frames):
        assert np.array_equal(planar[k], O.rgb_to_yuv420(rgb[k])), k
    command = scene.ffmpeg.command
    assert "yuv420p" in " ".join(str(part) for part in command)
    with pytest.raises(ValueError, match="pixel_format"):
        make(Visualizer, audio=(pcm, 44100), background=background).main(pixel_format="nv12", **kw)
