"""srt_render_noalloc (the entry point that takes its work memory from a fixed device array and allocates nothing)
against srt_render: bit-identical pixels, on two calls in a row of which the second is smaller, plus its refusals."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

srt = importlib.import_module("simple-ray-tracer_amd")
standin = importlib.import_module("simple-ray-tracer_amd.standin")


def _scene(name):
    import os
    from conftest import GOLD
    return standin.load_scene_npz(os.path.join(GOLD, "scenes", name + ".npz"))


def _both(bvh, cam, feat, nsx, nsy, rect, W, H, flags=0):
    import torch
    outs = []
    for call in (bvh.render_device, bvh.render_device_noalloc):
        buf = torch.full((H, W, 3), -7.0, dtype=torch.float32, device="cuda")
        p = bvh.params(cam, feat, nsx, nsy, 5, 3, rect, flags)
        call(p, buf.data_ptr(), W, H, 0, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(buf.cpu().numpy())
    return outs


def test_noalloc_equals_render_twice_in_a_row():
    import torch
    srt.set_device(0)
    monkey = srt.Bvh(srt.Scene.from_arrays(_scene("monkey")), srt.F_ACCEL)
    d = _scene("cornell_parallelogram")
    cornell = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)
    # first call: reflections (both path queues), hard shadows, 2 x 2 samples, ragged rectangle
    feat = (srt.F_ACCEL | srt.F_SHADING | srt.F_HARD_SHADOW | srt.F_RECURSIVE | srt.F_NORMAL_INTERP | srt.F_MULTISAMPLE | srt.F_MISS_CULL)
    W, H = 200, 150
    cam = srt.make_camera(W, H)
    want, got = _both(monkey, cam, feat, 2, 2, (3, 5, 197, 149), W, H)
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32)) and want.max() > 0.05
    assert (got[0, 0] == -7.0).all()  # outside the rectangle: untouched
    # second call, smaller, another scene and feature set (100 soft-shadow queries per hit): the scratch area still
    # holds the first call's queues and counters
    feat2 = srt.F_ACCEL | srt.F_SHADING | srt.F_SOFT_SHADOW | srt.F_MISS_CULL
    W2, H2 = 64, 48
    cam2 = srt.make_camera(W2, H2)
    want2, got2 = _both(cornell, cam2, feat2, 1, 1, (0, 0, W2, H2), W2, H2)
    assert np.array_equal(want2.view(np.uint32), got2.view(np.uint32)) and want2.max() > 0.05
    # and the first once more, with the counting kernels
    want3, got3 = _both(monkey, cam, feat, 2, 2, (3, 5, 197, 149), W, H, flags=srt.RENDER_STATS)
    assert np.array_equal(want3.view(np.uint32), got.view(np.uint32)) and np.array_equal(got3.view(np.uint32), got.view(np.uint32))


def test_noalloc_refusals():
    import torch
    srt.set_device(0)
    bvh = srt.Bvh(srt.Scene.from_arrays(_scene("monkey")), srt.F_ACCEL)
    feat = srt.F_ACCEL | srt.F_SHADING | srt.F_HARD_SHADOW | srt.F_RECURSIVE | srt.F_MISS_CULL
    W, H = 2048, 2048  # 4 Mi samples x 258 bytes: beyond the 64 MiB area
    buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    cam = srt.make_camera(W, H)
    stream = torch.cuda.current_stream().cuda_stream
    with pytest.raises(srt.SrtError, match="scratch area"):
        bvh.render_device_noalloc(bvh.params(cam, feat, 1, 1, 5, 0, None, 0), buf.data_ptr(), W, H, 0, 0, stream)
    # the same frame in rectangles that fit is fine and equals srt_render
    part = (512, 512, 960, 960)
    bvh.render_device_noalloc(bvh.params(cam, feat, 1, 1, 5, 0, part, 0), buf.data_ptr(), W, H, 0, 0, stream)
    ref = torch.zeros_like(buf)
    bvh.render_device(bvh.params(cam, feat, 1, 1, 5, 0, part, 0), ref.data_ptr(), W, H, 0, 0, stream)
    torch.cuda.synchronize()
    assert torch.equal(buf.view(torch.int32), ref.view(torch.int32)) and float(buf.max()) > 0.05
    with pytest.raises(srt.SrtError, match="transparency"):
        bvh.render_device_noalloc(bvh.params(cam, feat | srt.F_TRANSPARENCY, 1, 1, 5, 0, part, 0), buf.data_ptr(), W, H, 0, 0, stream)
    with pytest.raises(srt.SrtError, match="SRT_RENDER_TIMING"):
        bvh.render_device_noalloc(bvh.params(cam, feat, 1, 1, 5, 0, part, srt.RENDER_TIMING), buf.data_ptr(), W, H, 0, 0, stream)
    o = srt.Output()
    host = np.zeros((H, W, 3), np.float32)
    o.rgb = host.ctypes.data
    o.out_width, o.out_height, o.on_device = W, H, 0
    p = bvh.params(cam, feat, 1, 1, 5, 0, part, 0)
    import ctypes as C
    assert srt.host().srth_render_noalloc(bvh.h, C.byref(p), C.byref(o), None) != 0
    assert "device memory" in srt.host().srth_last_error().decode()


def test_noalloc_allocates_nothing():
    """What the entry point exists for: after one warm-up call, further calls leave the GPU's free memory exactly where it
    was (hipMemGetInfo through torch; srt_render with a larger rectangle right after does move it, which shows that the
    probe sees the library's allocations)."""
    import torch
    srt.set_device(0, context=7)  # a back-end handle of its own: its work buffers have not been grown by other tests
    bvh = srt.Bvh(srt.Scene.from_arrays(_scene("monkey")), srt.F_ACCEL)
    srt.set_device(0)
    feat = srt.F_ACCEL | srt.F_SHADING | srt.F_HARD_SHADOW | srt.F_RECURSIVE | srt.F_MULTISAMPLE | srt.F_MISS_CULL
    W, H = 256, 192
    cam = srt.make_camera(W, H)
    buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    big = torch.zeros((2160, 3840, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    p = bvh.params(cam, feat, 2, 2, 5, 0, None, 0)
    bvh.render_device_noalloc(p, buf.data_ptr(), W, H, 0, 0, stream)  # warm-up: module load, kernel code upload
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for rect in (None, (10, 10, 200, 150), (0, 0, 64, 64)):
        bvh.render_device_noalloc(bvh.params(cam, feat, 2, 2, 5, 0, rect, 0), buf.data_ptr(), W, H, 0, 0, stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    cam4k = srt.make_camera(3840, 2160)
    bvh.render_device(bvh.params(cam4k, feat, 2, 2, 5, 0, None, 0), big.data_ptr(), 3840, 2160, 0, 0, stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] < free0  # the allocating entry point grew the handle's work buffers

