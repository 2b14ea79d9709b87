import importlib, sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
srt = importlib.import_module("simple-ray-tracer_amd")
import bench
srt.set_device(0)
for wl in ("teapot", "monkey", "cube_textured"):
    d, cfg = bench.workload(wl, srt)
    bvh = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)
    W, H = cfg["width"], cfg["height"]
    cam = srt.make_camera(W, H)
    buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    # SRT_LATENCY_STREAM=pool: a stream of PyTorch's pool instead of the default stream (which hardware queue a stream
    # lands on decides whether the handle's second stream can run beside it)
    own = torch.cuda.Stream() if os.environ.get("SRT_LATENCY_STREAM") == "pool" else None
    if own is not None:
        torch.cuda.set_stream(own)
    st = torch.cuda.current_stream().cuda_stream
    for flags, what in ((0, "plain"), (srt.RENDER_TIMING, "timing events")):
        p = bvh.params(cam, cfg["features"], cfg["nsx"], cfg["nsy"], cfg["max_depth"], 0, None, flags)
        for _ in range(5):
            bvh.render_device(p, buf.data_ptr(), W, H, 0, 0, st)
        torch.cuda.synchronize()
        bvh.kernel_times(reset=True)
        t0 = time.perf_counter()
        N = 50
        for _ in range(N):
            bvh.render_device(p, buf.data_ptr(), W, H, 0, 0, st)
        torch.cuda.synchronize()
        print(wl, what, round(1e3 * (time.perf_counter() - t0) / N, 4), "ms/frame", flush=True)
        bvh.kernel_times(reset=True)
