#!/usr/bin/env python3
"""Throughput of the BVH-traversal / intersection / shading path on MI355X.

One step = one frame of the workload: every tile of the frame rendered by the HIP path (rays generated on the
device, scene already resident in HBM) and, with more than one GPU, the finished tiles gathered to rank 0 over
RCCL.  Ranks draw tiles from a shared counter (dynamic stealing); the frame is fixed, so more GPUs means strong
scaling.  A "ray" is one BVH query (camera, reflection or shadow), counted on the device.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints one JSON line with the whole-job Mrays/s, ms per frame, the roofline record of the dominant kernel
(device time from HIP events recorded around every launch of the timed steps) and, at N = 1, the CPU baseline
(the CPU restatement of the reference algorithm, reference visiting rule, timed on a bounded sample).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
L2_PEAK_GBS = 34500.0  # MI355X aggregate L2 bandwidth (8 XCDs x 4 MiB), the microarchitecture guide's figure


def workload(name, srt, dragon_obj=None):
    standin = importlib.import_module("simple-ray-tracer_amd.standin")
    if name in ("dragon", "dragon8k"):
        if dragon_obj:
            # the real model, through the host mirror's OBJ loader, centred and scaled to unit radius as the reference's
            # Dragon scene loads it (reference src/scene.cpp:70-75); lights as for the stand-in
            d = srt.Scene.from_obj(dragon_obj, normalize=True).to_arrays()
            what = f"{os.path.basename(dragon_obj)} ({len(d['tris'])} triangles, user supplied)"
        else:
            d = standin.dragon_standin(check=True)
            what = "dragon stand-in (procedural, 262144 triangles)"
        d["lights"] = standin.dragon_lights()
        feat = srt.F_ACCEL | srt.F_SHADING | srt.F_HARD_SHADOW | srt.F_SOFT_SHADOW | srt.F_MULTISAMPLE | srt.F_MISS_CULL | srt.F_COARSE_SHADOW
        if name == "dragon":
            return d, dict(width=3840, height=2160, features=feat, nsx=4, nsy=2, max_depth=5, cpu_block=1, tile=None,
                           label=what + " 3840x2160, 8 spp (4x2 grid), shading + "
                                 "hard shadow (point light) + soft shadow (parallelogram light, 100 samples)")
        # the stress frame: 2.12 G samples, every hit reflects (ks = 0.2) until the depth cap; a frame this size is
        # rendered in rectangles on one GPU too (the work memory of one rectangle is sized for its samples)
        return d, dict(width=7680, height=4320, features=feat | srt.F_RECURSIVE, nsx=8, nsy=8, max_depth=5, cpu_block=1, tile=(960, 540), lanes=1,
                       label=what + " 7680x4320, 64 spp (8x8 grid), shading + hard shadow "
                             "(point light) + soft shadow (parallelogram light, 100 samples) + mirror recursion (depth cap 5)")
    scene_file = os.path.join(ROOT, "tests", "golden", "scenes", name + ".npz")
    if name == "teapot":
        feat = srt.F_ACCEL | srt.F_SHADING | srt.F_HARD_SHADOW | srt.F_MISS_CULL | srt.F_COARSE_SHADOW
        return standin.load_scene_npz(scene_file), dict(width=1920, height=1080, features=feat, nsx=1, nsy=1, max_depth=5, cpu_block=4, tile=None,
                                                        label="teapot 15704 triangles 1920x1080, primary + shadow rays")
    if name in ("monkey", "cube_textured"):
        feat = (srt.F_ACCEL | srt.F_SHADING | srt.F_HARD_SHADOW | srt.F_RECURSIVE | srt.F_NORMAL_INTERP | srt.F_TEXTURE
                | srt.F_MISS_CULL | srt.F_COARSE_SHADOW)
        return standin.load_scene_npz(scene_file), dict(width=1920, height=1080, features=feat, nsx=1, nsy=1, max_depth=5, cpu_block=8, tile=None,
                                                        label=f"{name} 1920x1080, shading + hard shadows + mirror recursion (depth cap 5) + "
                                                              "normal interpolation + texture lookup")
    raise SystemExit(f"unknown workload {name}")


def algorithmic_bytes(c, node_bytes=64, tri_bytes=36):
    """node_bytes per inner-node visit (64: one node record, whether it holds the two child boxes of a binary node or
    the four 16-bit boxes of a wide one) + tri_bytes per triangle test (36: the three vertex positions; 96: the whole
    traversal record) + 48 B of ray I/O per query + 64 B of attributes per shaded hit."""
    return node_bytes * c["inner_visits"] + tri_bytes * c["tri_tests"] + 48 * (c["path_rays"] + c["shadow_rays"]) + 64 * c["shaded_hits"]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_sample(d, cfg, seconds, threads, features, block, max_blocks, rule):
    """The CPU restatement on seeded random pixel blocks of the same frame, under the given traversal policy.
    Worker threads draw blocks from one shared list until the list or the time budget is used up (a block that holds
    hits costs orders of magnitude more than one that misses, so blocks are drawn dynamically).  The spread is
    estimated from eight interleaved groups of the drawn blocks (group throughput = threads x rays / CPU seconds)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import threading
    import orc
    bvh = orc.Bvh(d, orc.F_ACCEL)
    cam = orc.make_camera(cfg["width"], cfg["height"])
    rng = np.random.default_rng(2026)
    bw = block
    nbx, nby = cfg["width"] // bw, cfg["height"] // bw
    order = rng.permutation(nbx * nby)[:max_blocks]
    it = iter(enumerate(order))
    lock = threading.Lock()
    t0 = time.perf_counter()
    deadline = t0 + seconds
    groups = 8
    g_rays, g_secs = [0] * groups, [0.0] * groups
    totals = [0, 0, 0, 0.0]

    def worker():
        rays = blocks = hit_blocks = 0
        busy = 0.0
        mine = [[0, 0.0] for _ in range(groups)]
        while time.perf_counter() < deadline:
            with lock:
                nb = next(it, None)
            if nb is None:
                break
            k, b = nb
            x, y = int(b % nbx) * bw, int(b // nbx) * bw
            ta = time.perf_counter()
            r = bvh.render(cam, features, nsx=cfg["nsx"], nsy=cfg["nsy"], max_depth=cfg["max_depth"], rect=(x, y, x + bw, y + bw),
                           threads=1, primary=False)
            tb = time.perf_counter()
            c = r["counters"]
            n = c["path_rays"] + c["shadow_rays"]
            rays += n
            blocks += 1
            busy += tb - ta
            hit_blocks += 1 if c["shaded_hits"] else 0
            mine[k % groups][0] += n
            mine[k % groups][1] += tb - ta
        with lock:
            totals[0] += rays
            totals[1] += blocks
            totals[2] += hit_blocks
            totals[3] += busy
            for g in range(groups):
                g_rays[g] += mine[g][0]
                g_secs[g] += mine[g][1]

    ts = [threading.Thread(target=worker) for _ in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    rays, blocks, hit_blocks, busy = totals
    rates = [threads * g_rays[g] / g_secs[g] / 1e6 for g in range(groups) if g_secs[g] > 0]
    stderr = float(np.std(rates, ddof=1) / np.sqrt(len(rates))) if len(rates) > 1 else None
    # all threads busy until the list or the budget ran out: rays per CPU second times the threads (when the list runs
    # out, the last blocks leave threads idle; wall-clock time would then understate what the cores can do)
    value = threads * rays / busy / 1e6 if busy > 0 else 0.0
    return {"value": value, "unit": "Mrays/s", "cores": threads, "kind": "port", "stderr": stderr,
            "sample": f"{blocks} seeded random {bw}x{bw}-pixel blocks of the same frame, {hit_blocks} of them with hits "
                      f"({rays} rays in {dt:.1f} s wall, {busy:.1f} CPU s; standard error over {len(rates)} interleaved groups of blocks "
                      f"{'%.2g' % stderr if stderr is not None else 'n/a'} Mrays/s), {rule}, {threads} threads"}


def cpu_baseline(d, cfg, seconds, threads):
    """Two CPU figures beside the GPU's: the restatement under the REFERENCE's visiting rule (what the reference itself
    does: an unoccluded shadow ray walks the whole tree) and under the policy the GPU runs (miss culling, shadow
    queries on coarse boxes), on seeded pixel blocks of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    ref_feat = cfg["features"] & ~(orc.F_MISS_CULL | orc.F_COARSE_SHADOW)
    out = cpu_sample(d, cfg, seconds, threads, ref_feat, cfg["cpu_block"], 1 << 30, "reference visiting rule (no miss culling)")
    if cfg["cpu_block"] == 1:
        out["sample"] += ("; blocks are single pixels because one hit pixel already costs seconds under the reference rule "
                          "(64 tiles of 32x32 pixels would take days)")
    out["cpu_model"] = cpu_model()
    same = cpu_sample(d, cfg, max(seconds, 10.0), threads, cfg["features"], 32, cfg.get("cpu_tiles", 512),
                      "the GPU's traversal policy (miss culling, shadow queries on coarse four-wide boxes)")
    out["same_policy"] = {k: same[k] for k in ("value", "unit", "cores", "stderr", "sample")}
    return out


PROFILE_KERNELS = {"trace_shadow": "void k_trace_shadow<true, false, true>", "trace_primary": "void k_trace_primary<true, false>",
                   "trace_secondary": "void k_trace_secondary<true, false>"}
KERNEL_SOURCES = ("simple-ray-tracer_amd/csrc/srt_kernels.hip", "simple-ray-tracer_amd/csrc/srt_device.h", "include/srt_hip.h")


def kernel_source_hash():
    """SHA-256 over the sources libsrt_hip.so is built from: tells which kernels a profile summary belongs to."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


def git_blob_hash(path):
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def profiled_counters(family, summary, label):
    """Hardware counters of one kernel family from a rocprofv3 --pmc summary (profiles/summarize.py) of this same
    command: fabric-side traffic per launch and the cache side of the kernel.  They cannot be read from inside a
    run.  The summary counts only if it was taken with the kernels now in the tree (its recorded source hash) on
    this workload; otherwise the record says why it was not used."""
    if summary is None:
        cands = sorted((os.path.join(ROOT, "profiles", f) for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_summary.txt")),
                       reverse=True)
    else:
        cands = [summary if os.path.isabs(summary) else os.path.join(ROOT, summary)]
    want_hash = kernel_source_hash()
    why = "no profile summary given or found"
    for path in cands:
        if not os.path.exists(path):
            why = f"{path} does not exist"
            continue
        head = {}
        for line in open(path):
            if line.startswith("# ") and ": " in line:
                key, val = line[2:].split(": ", 1)
                head[key.strip()] = val.strip()
        rel = os.path.relpath(path, ROOT)
        if head.get("kernel-source-sha256") != want_hash:
            why = f"{rel} was taken with other kernel sources"
            continue
        if head.get("workload") != label:
            why = f"{rel} was taken on another workload"
            continue
        name = PROFILE_KERNELS.get(family)
        c = {}
        for line in open(path):
            if name and line.startswith(name) and "per_launch=" in line:
                f = line[len(name):].split()
                c[f[0]] = float(f[-1].split("=")[1])
        need = ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "TCP_TOTAL_CACHE_ACCESSES_sum", "TA_TA_BUSY_sum", "TCC_HIT_sum",
                "TCC_MISS_sum", "SQ_INSTS_VALU", "SQ_INSTS_SALU")
        if any(k not in c for k in need):
            why = f"{rel} does not cover kernel family {family}"
            continue
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # the counter is summed over the 8 XCDs
        n_cu, n_simd = 256, 1024
        return {
            # FETCH_SIZE and WRITE_SIZE are in KiB and count the L2's fabric-side requests, Infinity Cache hits included
            # (the 33 MB of records exceed one XCD's 4 MB L2 but stay in the 256 MB Infinity Cache); no correction
            # applied: the x2 of wide streaming reads does not apply to 16-byte gathers and byte stores
            "traffic": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
            "cache_side": {
                "l1_tag_lookups_per_cycle_per_cu": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / n_cu / cycles,
                "texture_addresser_busy": c["TA_TA_BUSY_sum"] / n_cu / cycles,
                "l2_hit": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
                "valu_pipe_busy": c["SQ_INSTS_VALU"] * 2.0 / n_simd / cycles,  # a wave64 VALU instruction holds its SIMD 2 cycles
                "salu_per_valu": c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"],
            },
            "profile_summary": {"path": rel, "git_blob": git_blob_hash(path), "kernel_source_sha256": want_hash},
        }
    return {"traffic": None, "profile_summary": None, "traffic_note": why}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragon", choices=["dragon", "dragon8k", "teapot", "monkey", "cube_textured"])
    ap.add_argument("--tile", type=int, nargs=2, default=None, help="tile width height (default: whole frame on 1 GPU; a 4 x 4, 8 x 4 or 8 x 8 grid of tiles for 2, up to 4, more ranks; 960x540 for dragon8k)")
    ap.add_argument("--profile-summary", default=None, help="rocprofv3 --pmc summary (profiles/summarize.py output) to take the fabric-side "
                    "traffic and cache counters from; default: the newest profiles/*_summary.txt taken with the current kernel sources")
    ap.add_argument("--dragon-obj", default=None, help="OBJ file of the real dragon (absent from the reference checkout) to render instead of the "
                    "procedural stand-in; loaded, centred and scaled as the reference's Dragon scene")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact-shadow-boxes", action="store_true", help="shadow queries on the full-precision node array (F_MISS_CULL alone) instead of the coarse one")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--gather-probe", action="store_true", help="with --gpus 1: cut the frame into 480x270 tiles and run the many-rank frame path "
                    "alone (one-rank RCCL group, tile counter and frame board in shared memory), timing the parts of the gather: what one GPU can "
                    "tell about the fixed per-frame cost of N ranks")
    ap.add_argument("--lanes", type=int, default=None, help="back-end contexts (each with its stream) a tiled frame alternates between "
                    "(default 2; 1 for dragon8k, whose rectangles are large enough for their tails not to matter)")
    ap.add_argument("--in-flight", type=int, default=None, help="tiles a rank may have enqueued and unfinished (default: one per lane)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible: this benchmark measures the HIP path and has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 and not (hasattr(dist, "batch_isend_irecv") and hasattr(dist, "P2POp")):
        raise SystemExit("this torch.distributed has no batch_isend_irecv / P2POp: the tile gather to rank 0 needs them (PyTorch >= 1.8)")
    probe = args.gather_probe and world == 1
    if probe:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    srt = importlib.import_module("simple-ray-tracer_amd")
    srt.set_device(local_rank)
    d, cfg = workload(args.workload, srt, args.dragon_obj)
    if args.exact_shadow_boxes:
        cfg["features"] &= ~srt.F_COARSE_SHADOW
        PROFILE_KERNELS["trace_shadow"] = "void k_trace_shadow<true, false, false>"
    W, H = cfg["width"], cfg["height"]
    t_build = time.perf_counter()
    bvh = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)
    t_build = time.perf_counter() - t_build
    cam = srt.make_camera(W, H)

    tiling = importlib.import_module("simple-ray-tracer_amd.distributed")
    # Tiles per rank: enough of them for stealing to even out the ranks (about eight each), no more, because a frame in 64
    # tiles costs 7 % more GPU time than the same frame in one piece and one in 16 tiles 2 % (4K dragon frame, one GPU).
    by_world = (W, H) if (world == 1 and not probe) else ((W // 4, H // 4) if world <= 2 and not probe else ((W // 8, H // 4) if world <= 4 and not probe else (W // 8, H // 8)))
    tw, th = args.tile if args.tile else (cfg["tile"] if cfg["tile"] else by_world)
    counter = board = None
    if world > 1 or probe:
        name = f"/srt_tiles_{os.environ.get('MASTER_PORT', '0')}"
        n_rects = len(tiling.tile_rects(W, H, tw, th))
        if rank == 0:
            counter = srt.TileCounter(name, True)
            board = srt.FrameBoard(name + "_board", True, world, n_rects)
        dist.barrier()
        if rank != 0:
            counter = srt.TileCounter(name, False)
            board = srt.FrameBoard(name + "_board", False, world, n_rects)
    n_lanes = max(1, args.lanes if args.lanes is not None else cfg.get("lanes", 2))
    tiler = tiling.FrameTiler(W, H, tw, th, dev, rank, world, counter, board=board, max_in_flight=args.in_flight if args.in_flight else max(2, n_lanes))
    gather_parts = []
    n_tiles = tiler.n_tiles
    # A frame cut into tiles alternates between two back-end contexts on two streams, so that the tail of one
    # tile's kernels overlaps the start of the next tile's (a context owns its work buffers and scene copy).
    lanes = [(bvh, torch.cuda.current_stream())]
    if n_tiles > 1:
        # every lane on a stream of its own: launches on the default stream were measured not to run beside those of
        # another lane at all (64 tiles: 125.1 ms per frame with lane 0 on the default stream, 118.5 ms without)
        lanes = [(bvh, torch.cuda.Stream())]
        for ctx in range(1, n_lanes):
            srt.set_device(local_rank, context=ctx)
            lanes.append((srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL), torch.cuda.Stream()))
        srt.set_device(local_rank, context=0)

    def lane():
        return lanes[tiler.current % len(lanes)]

    def stream_fence():
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(lane()[1])
        return ev

    def one_step(flags, features=None):
        """One frame: this rank's share of the tiles (enqueued on the lanes' streams, written straight into the
        rank's tile buffer in HBM), then the gather to rank 0."""
        feat = cfg["features"] if features is None else features

        def render_tile(rect, slot):
            b, stream = lane()
            p = b.params(cam, feat, cfg["nsx"], cfg["nsy"], cfg["max_depth"], 0, rect, flags)
            b.render_device(p, slot.data_ptr(), tw, th, rect[0], rect[1], stream.cuda_stream)

        t_a = time.perf_counter()
        k = tiler.render(render_tile, stream_fence if n_tiles > 1 else None)
        if os.environ.get("SRT_BENCH_DEBUG"):
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            tiler.gather(k)
            torch.cuda.synchronize()
            print(f"[rank {rank}] tiles {k} render {1e3 * (t_b - t_a):.1f} ms gather {1e3 * (time.perf_counter() - t_b):.1f} ms", file=sys.stderr, flush=True)
        else:
            tiler.gather(k)
        gather_parts.append(dict(tiler.gather_ms))
        return k

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def summed(parts):
        """Element-wise sum of the lanes' counter dictionaries (one level of nesting allowed)."""
        out = {}
        for part in parts:
            for key, val in part.items():
                if isinstance(val, dict):
                    out.setdefault(key, {})
                    for k2, v2 in val.items():
                        out[key][k2] = out[key].get(k2, 0) + v2
                else:
                    out[key] = out.get(key, 0) + val
        return out

    def read_counters():
        return summed([b.read_counters(reset=True) for b, _ in lanes])

    def kernel_times():
        return summed([b.kernel_times(reset=True) for b, _ in lanes])

    def read_kernel_counters():
        return summed([b.read_kernel_counters(reset=True) for b, _ in lanes])

    for _ in range(args.warmup):
        one_step(0)
    fence()
    read_counters()
    kernel_times()
    gather_parts.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(srt.RENDER_TIMING)
    fence()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    c = read_counters()
    rays = torch.tensor([c["path_rays"] + c["shadow_rays"]], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    rays_per_frame = float(rays.item()) / args.steps
    times = kernel_times()
    timed_gather_parts = list(gather_parts)
    # The timed steps carry a pair of HIP events around every launch (that is where avg_launch_ms comes from).  On a frame
    # of a millisecond the pairs are a visible part of the frame, so the same K steps run once more without them.
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        one_step(0)
    fence()
    plain = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(plain, op=dist.ReduceOp.MAX)
    ms_plain = float(plain.item()) / args.steps * 1e3
    read_counters()
    gather_parts[:] = timed_gather_parts
    gather_ms = {key: sum(g[key] for g in gather_parts) / max(len(gather_parts), 1) for key in (gather_parts[0] if gather_parts else {})}
    if world > 1 or probe:
        tb = []
        for _ in range(5):
            ta = time.perf_counter()
            dist.barrier()
            torch.cuda.synchronize()
            tb.append(1e3 * (time.perf_counter() - ta))
        gather_ms["barrier"] = sorted(tb)[len(tb) // 2]

    # work of each kernel family, from one more (untimed) frame with the counting kernels ...
    read_kernel_counters()
    one_step(srt.RENDER_STATS)
    fence()
    fam = read_kernel_counters()
    # ... and the same frame once more under plain miss culling on the binary tree's full-precision boxes: the visits the
    # ALGORITHM asks for (one 64-byte node per inner visit, level by level), whatever node encoding the timed kernels walk
    fam_s1 = fam
    if cfg["features"] & srt.F_COARSE_SHADOW:
        one_step(srt.RENDER_STATS, cfg["features"] & ~srt.F_COARSE_SHADOW)
        fence()
        fam_s1 = read_kernel_counters()

    if rank == 0:
        dominant = max(times, key=lambda n: times[n]["ms"])
        launches_timed = max(times[dominant]["launches"], 1)
        avg_ms = times[dominant]["ms"] / launches_timed
        launches_per_frame = launches_timed / args.steps
        secs = {n: times[n]["ms"] / args.steps * 1e-3 for n in times}

        def rate(counters, n, **kw):
            return algorithmic_bytes(counters[n], **kw) / secs[n] / 1e9 if secs[n] > 0 else 0.0

        # Three ways to price the dominant kernel's launch, all against FIXED peaks (the guide's aggregate L2 bandwidth;
        # HBM for the traffic that leaves the caches):
        #   achieved          algorithmic bytes: 64 B per inner-node visit and 36 B per triangle test of the level-by-level
        #                     walk under miss culling (counted by the extra frame above) + 48 B per query + 64 B per shaded
        #                     hit, over the launch time of the kernel as it runs (which may take fewer, wider visits)
        #   as_walked         the same formula on the visits the timed kernel really makes (a wide record is 64 B too)
        #   requested         bytes its loads ask the caches for: 64 B per visit, the whole 96-byte record per triangle test
        achieved = rate(fam_s1, dominant)
        as_walked = rate(fam, dominant)
        requested = rate(fam, dominant, tri_bytes=96)
        bytes_per_launch = algorithmic_bytes(fam_s1[dominant]) / max(launches_per_frame, 1e-9)
        # Context, not a peak: what this GPU serves when lanes gather 64-byte records of the same arrays with no
        # arithmetic in between, from a 2 MiB window (stays in every XCD's L2) and from all records (a walk without
        # locality; beyond one XCD's L2, served by the Infinity Cache).
        probe_l2, l2_window = bvh.probe_gather(2 << 20, 512)
        ws_rate, ws_bytes = bvh.probe_gather(0, 512)
        out = {
            "metric": "Mrays/s (and ms/frame as ms_per_step), dragon 3840x2160 8spp" if args.workload == "dragon" else "Mrays/s (and ms/frame as ms_per_step)",
            "value": rays_per_frame * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_without_launch_events": ms_plain,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if not args.dragon_obj else "user-supplied model, synthetic camera and lights",
            "config": {"workload": cfg["label"], "tile": [tw, th], "tiles": n_tiles,
                       "traversal": "miss-cull" + (", shadow queries on coarse four-wide nodes" if cfg["features"] & srt.F_COARSE_SHADOW else ""),
                       "rays_per_frame": rays_per_frame, "bvh_levels": bvh.num_levels, "bvh_build_upload_s": round(t_build, 3)},
            "roofline": {"bound": "l2", "kernel": dominant, "achieved": achieved, "peak": L2_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / L2_PEAK_GBS, "traffic": None,
                         "peak_source": "aggregate L2 bandwidth of the microarchitecture guide (the records never leave the caches: see traffic / hbm_frac)",
                         "frac_l2": achieved / L2_PEAK_GBS,
                         "frac_l2_as_walked": as_walked / L2_PEAK_GBS,
                         "frac_l2_bytes_requested": requested / L2_PEAK_GBS,
                         "achieved_as_walked": as_walked, "achieved_bytes_requested": requested,
                         "gather_probe": {"l2_window_rate": probe_l2, "l2_window_bytes": l2_window, "all_records_rate": ws_rate, "all_records_bytes": ws_bytes,
                                          "unit": "GB/s", "note": "64-byte record gathers with no arithmetic, this GPU, this scene: context, not the peak"},
                         "avg_launch_ms": avg_ms, "launches_timed": launches_timed,
                         "concurrent_lanes": len(lanes),  # > 1: launches of different tiles overlap, so do their timed intervals
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "kernel_ms_per_frame": {n: times[n]["ms"] / args.steps for n in times},
                         # the same ratios for every traversal family (per frame: bytes over summed launch time)
                         "by_kernel": {n: {"achieved": rate(fam_s1, n), "frac": rate(fam_s1, n) / L2_PEAK_GBS, "frac_as_walked": rate(fam, n) / L2_PEAK_GBS}
                                       for n in times if n.startswith("trace_") and secs[n] > 0},
                         "work_per_frame": {n: fam[n] for n in fam if fam[n]["path_rays"] + fam[n]["shadow_rays"] + fam[n]["shaded_hits"]},
                         "level_by_level_work_per_frame": {n: {k: fam_s1[n][k] for k in ("inner_visits", "tri_tests")}
                                                           for n in fam_s1 if n.startswith("trace_") and fam_s1[n]["inner_visits"]}},
        }
        if world == 1:
            out["roofline"].update(profiled_counters(dominant, args.profile_summary, cfg["label"]))
            tr = out["roofline"].get("traffic")
            out["roofline"]["hbm_frac"] = tr / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if tr and avg_ms > 0 else None
        if gather_ms:
            # per frame, rank 0's host wall clock: wait_tiles = until this rank's own tiles are done (rendering, not a cost of
            # the gather); ids = posting and collecting the tile lists through shared memory; pixels = placing own tiles and
            # receiving the peers'; settle = the stream synchronisation that frees the tile buffers; barrier = one
            # dist.barrier() + device synchronisation (paid once per run, between the timed steps and the clock)
            out["gather_ms"] = dict(gather_ms, world=world, tiles=n_tiles,
                                    fixed_cost=sum(v for k2, v in gather_ms.items() if k2 in ("ids", "pixels", "settle")))
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(d, cfg, args.cpu_seconds, threads)
            cb = out["cpu_baseline"]
            cb["gpu_over_cpu"] = {"reference_rule": out["value"] / cb["value"] if cb["value"] > 0 else None,
                                  "same_policy": out["value"] / cb["same_policy"]["value"] if cb["same_policy"]["value"] > 0 else None}
        print(json.dumps(out), flush=True)
    if counter:
        if world > 1:
            dist.barrier()
        counter.close()
        board.close()
    if world > 1 or probe:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
