"""How long does sfx_rgb_to_yuv420 take beside the render it follows? (HIP events on the context's stream; 4K and 1080p frames)
    python tools/experiments/yuv_kernel_time.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from shaderflow_amd import _native as N

context = N.Context(0, None)
for width, height, frames in ((3840, 2160, 75), (3840, 2160, 300), (1920, 1080, 300)):
    rgb = context.alloc(width*height*3*frames)
    yuv = context.alloc(width*height*3//2*frames)
    for _ in range(2):
        context.rgb_to_yuv420(rgb, yuv, width, height, frames)
    context.synchronize()
    context.event_record(0)
    for _ in range(5):
        context.rgb_to_yuv420(rgb, yuv, width, height, frames)
    context.event_record(1)
    ms = context.event_elapsed_ms(0, 1)/5
    moved = width*height*4.5*frames
    print(f"{width}x{height} x {frames} frames: {ms*1e3/frames:.2f} us per frame, {moved/ms/1e6:.0f} GB/s read + written")
    context.free(rgb); context.free(yuv)
