"""Does replaying a frame's launches as one HIP graph help a launch-bound frame?  srt_render_noalloc (allocation free,
stream ordered) is captured on a torch stream and replayed; compared with calling it directly.  usage (GPU box):
python3 tools/graph_probe.py
Run it in a process of its own.  The same capture at the end of a test process that had rendered with several handles
before ended with a GPU fault whose cause was not established (capture and replay are PyTorch's and the HIP runtime's;
the library only sees stream-ordered calls), which is why no test captures a graph."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
srt = importlib.import_module("simple-ray-tracer_amd")
import bench
srt.set_device(0)
for wl, (W, H) in (("cube_textured", (480, 360)), ("monkey", (480, 360))):
    d, cfg = bench.workload(wl, srt)
    bvh = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)
    cam = srt.make_camera(W, H)
    buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    ref = torch.zeros_like(buf)
    p = bvh.params(cam, cfg["features"], 1, 1, cfg["max_depth"], 0, None, 0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            bvh.render_device_noalloc(p, ref.data_ptr(), W, H, 0, 0, s.cuda_stream)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            bvh.render_device_noalloc(p, ref.data_ptr(), W, H, 0, 0, s.cuda_stream)
        s.synchronize()
        direct = 1e3 * (time.perf_counter() - t0) / 100
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        bvh.render_device_noalloc(p, buf.data_ptr(), W, H, 0, 0, torch.cuda.current_stream().cuda_stream)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        g.replay()
    torch.cuda.synchronize()
    graph = 1e3 * (time.perf_counter() - t0) / 100
    same = torch.equal(buf.view(torch.int32), ref.view(torch.int32))
    print(f"{wl} {W}x{H}: direct {direct:.4f} ms/frame, graph replay {graph:.4f} ms/frame, identical pixels {same}", flush=True)
