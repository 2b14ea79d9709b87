"""Host side of the product (no GPU): BVH build and flattening against the oracle, the OBJ/PNG loader against
the reference loader's recorded output, and the exported C ABI."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

from conftest import GOLD, assert_bits_equal

srt = importlib.import_module("simple-ray-tracer_amd")

SCENES = ["single_triangle", "cube", "cube_textured", "cornell", "cornell_parallelogram", "monkey", "teapot", "custom"]
REF_DATA = "/root/reference/data"
# SceneType values, reference src/scene.h:15-26
SCENE_TYPE = {"single_triangle": 0, "cube": 1, "cube_textured": 2, "cornell": 3, "cornell_parallelogram": 4,
              "monkey": 5, "teapot": 6, "custom": 9}


def test_abi_library_exports_every_declared_symbol():
    lib = C.CDLL(srt.HIP_LIB)
    header = open(os.path.join(os.path.dirname(srt.HERE), "include", "srt_hip.h")).read()
    import re
    declared = set(re.findall(r"\b(srt_[a-z_]+)\s*\(", header))
    assert declared == set(srt.HIP_ABI_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("sah", [False, True])
def test_bvh_build_matches_oracle(orc, scenes, name, sah):
    d = scenes(name)
    feat = orc.F_ACCEL | (orc.F_SAH if sah else 0)
    ob = orc.Bvh(d, feat)
    hb = srt.HostBvh(srt.Scene.from_arrays(d), srt.F_ACCEL | (srt.F_SAH if sah else 0))
    assert (hb.num_nodes, hb.num_levels, hb.num_leaves) == (ob.num_nodes, ob.num_levels, ob.num_leaves)
    om, obox = ob.dump()
    hm, hbox = hb.dump()
    assert np.array_equal(om, hm)
    assert_bits_equal(obox, hbox, "node boxes")
    assert np.array_equal(ob.tri_order(), hb.tri_order())


def _grid_scene(n):
    """Two meshes of n x n axis-aligned quads on a regular lattice: thousands of triangles with EQUAL centroid
    coordinates, so the build's result hangs on std::sort's order of equal keys at every level."""
    standin = importlib.import_module("simple-ray-tracer_amd.standin")
    d = standin.dragon_standin(4, 4)
    g = np.arange(n + 1, dtype=np.float32) / np.float32(n) - np.float32(0.5)
    xs, zs = np.meshgrid(g, g, indexing="ij")
    verts, tris = [], []
    for m, y in enumerate((np.float32(-0.25), np.float32(0.25))):
        v = np.zeros(((n + 1) * (n + 1), 8), np.float32)
        v[:, 0], v[:, 1], v[:, 2], v[:, 4] = xs.ravel(), y, zs.ravel(), 1.0
        i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        a = (i * (n + 1) + j).ravel()
        t = np.empty((2 * n * n, 3), np.uint32)
        t[0::2] = np.stack([a, a + 1, a + n + 2], 1)
        t[1::2] = np.stack([a, a + n + 2, a + n + 1], 1)
        verts.append(v), tris.append(t)
    d["verts"], d["tris"] = np.concatenate(verts), np.concatenate(tris)
    d["mesh_vert_off"] = np.array([0, len(verts[0]), len(d["verts"])], np.uint32)
    d["mesh_tri_off"] = np.array([0, len(tris[0]), len(d["tris"])], np.uint32)
    d["materials"] = np.repeat(d["materials"], 2, axis=0)
    d["mesh_tex"] = np.array([-1, -1], np.int32)
    return d


def test_dragon_standin_is_pinned():
    """The procedural stand-in for the missing dragon.obj is built with float64 numpy sin / cos before it is rounded
    to float32: every bench number and every full-size property test assumes these exact arrays."""
    import hashlib
    standin = importlib.import_module("simple-ray-tracer_amd.standin")
    d = standin.dragon_standin()
    assert d["verts"].shape == (131072, 8) and d["tris"].shape == (262144, 3)
    assert hashlib.sha256(np.ascontiguousarray(d["verts"]).tobytes()).hexdigest() == standin.DRAGON_STANDIN_SHA256["verts"]
    assert hashlib.sha256(np.ascontiguousarray(d["tris"]).tobytes()).hexdigest() == standin.DRAGON_STANDIN_SHA256["tris"]


@pytest.mark.parametrize("which", ["dragon_standin", "lattice"])
@pytest.mark.parametrize("sah", [False, True])
def test_large_bvh_build_matches_oracle(orc, which, sah):
    """Scenes large enough for the builder to fork its top levels over threads: same node table as the oracle's
    plain recursion, including the depth-capped leaves (stand-in) and massive key ties (lattice)."""
    standin = importlib.import_module("simple-ray-tracer_amd.standin")
    d = standin.dragon_standin() if which == "dragon_standin" else _grid_scene(48)
    ob = orc.Bvh(d, orc.F_ACCEL | (orc.F_SAH if sah else 0))
    hb = srt.HostBvh(srt.Scene.from_arrays(d), srt.F_ACCEL | (srt.F_SAH if sah else 0))
    assert (hb.num_nodes, hb.num_levels, hb.num_leaves) == (ob.num_nodes, ob.num_levels, ob.num_leaves)
    om, obox = ob.dump()
    hm, hbox = hb.dump()
    assert np.array_equal(om, hm)
    assert_bits_equal(obox, hbox, "node boxes")
    assert np.array_equal(ob.tri_order(), hb.tri_order())


@pytest.mark.parametrize("name", ["cube", "cornell", "monkey", "teapot"])
@pytest.mark.parametrize("bfs", [0, 4])
def test_flatten(orc, scenes, name, bfs):
    d = scenes(name)
    hb = srt.HostBvh(srt.Scene.from_arrays(d))
    f = hb.flatten(bfs)
    meta, box = hb.dump()
    n_inner = int((meta[:, 0] == 0).sum())
    assert len(f["nodes"]) == n_inner and len(f["tris"]) == len(d["tris"])
    # walk the flattened tree and the reference-order table side by side
    order = hb.tri_order()
    leaf_first = {}
    for i, m in enumerate(meta):
        if m[0]:
            leaf_first[i] = (m[2], m[3])

    def check(ref_idx, flat_ref):
        m = meta[ref_idx]
        if m[0]:
            assert flat_ref & 0x80000000
            assert (flat_ref & 0xFFFFFF, (flat_ref >> 24) & 0x7F) == (m[2], m[3])
            return
        assert not (flat_ref & 0x80000000)
        nd = f["nodes"][flat_ref]
        assert_bits_equal(nd["lbox"], box[m[2]], "left box")
        assert_bits_equal(nd["rbox"], box[m[3]], "right box")
        for child, ref in ((m[2], nd["left"]), (m[3], nd["right"])):
            if not meta[child][0]:
                assert ref > flat_ref  # children are stored after their parent
            check(child, int(ref))

    import sys
    sys.setrecursionlimit(10000)
    check(len(meta) - 1, f["root_ref"])
    assert_bits_equal(f["root_box"], box[-1], "root box")
    # triangle records: positions, ids and scene-order permutation
    verts, tris = d["verts"], d["tris"]
    for k in (0, len(order) // 2, len(order) - 1):
        mesh, ti = order[k]
        vi = tris[d["mesh_tri_off"][mesh] + ti] + d["mesh_vert_off"][mesh]
        assert_bits_equal(f["tris"][k]["v0"], verts[vi[0], :3], "v0")
        assert_bits_equal(f["tris"][k]["v2"], verts[vi[2], :3], "v2")
        assert (f["tris"][k]["mesh"], f["tris"][k]["tri"]) == (mesh, ti)
        assert f["scene_order"][d["mesh_tri_off"][mesh] + ti] == k
        assert_bits_equal(f["attrs"][k]["n1"], verts[vi[1], 3:6], "n1")
    # precomputed plane: unit normal and D = dot(v0, n) in float32, left to right
    t = f["tris"]
    e1, e2 = t["v1"] - t["v0"], t["v2"] - t["v0"]
    c = np.stack([e1[:, 1] * e2[:, 2] - e2[:, 1] * e1[:, 2], e1[:, 2] * e2[:, 0] - e2[:, 2] * e1[:, 0],
                  e1[:, 0] * e2[:, 1] - e2[:, 0] * e1[:, 1]], 1).astype(np.float32)
    assert_bits_equal(t["c2"], c, "c2 = cross(v1 - v0, v2 - v0)")
    c1 = np.stack([e2[:, 1] * e1[:, 2] - e1[:, 1] * e2[:, 2], e2[:, 2] * e1[:, 0] - e1[:, 2] * e2[:, 0],
                   e2[:, 0] * e1[:, 1] - e1[:, 0] * e2[:, 1]], 1).astype(np.float32)
    assert_bits_equal(t["c1"], c1, "c1 = cross(v2 - v0, v1 - v0)")
    dd = ((c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]) + c[:, 2] * c[:, 2]).astype(np.float32)
    n = c * (np.float32(1.0) / np.sqrt(dd))[:, None]
    assert_bits_equal(t["n"], n, "plane normal")
    D = ((t["v0"][:, 0] * n[:, 0] + t["v0"][:, 1] * n[:, 1]) + t["v0"][:, 2] * n[:, 2]).astype(np.float32)
    assert_bits_equal(t["D"], D, "plane D")


def _same_scene(a, b):
    for k in ("mesh_vert_off", "mesh_tri_off", "tris", "mesh_tex", "tex_w", "tex_h"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("verts", "materials", "texels", "lights"):
        assert_bits_equal(a[k], b[k], k)


@pytest.mark.parametrize("case", ["quads_groups", "polygon"])
@pytest.mark.parametrize("normalize", [0, 1])
def test_obj_loader_authored_cases(orc, case, normalize):
    want = orc.load_scene_npz(os.path.join(GOLD, "obj_cases", f"{case}.norm{normalize}.npz"))
    got = srt.Scene.from_obj(os.path.join(GOLD, "obj_cases", case + ".obj"), bool(normalize)).to_arrays()
    want["lights"] = want["lights"].reshape(-1, srt.LIGHT_STRIDE)
    _same_scene(got, want)


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="reference data directory not present on this machine")
@pytest.mark.parametrize("name", SCENES)
def test_prebuilt_scenes_match_reference_loader(scenes, name):
    got = srt.Scene.prebuilt(SCENE_TYPE[name], REF_DATA).to_arrays()
    _same_scene(got, scenes(name))


IMAGE_CASES = sorted(f for f in os.listdir(os.path.join(GOLD, "image_cases")) if f.endswith((".jpg", ".png")))


@pytest.mark.parametrize("name", IMAGE_CASES)
def test_image_loader_matches_reference_decoder(name):
    """Image(path): baseline / progressive JPEG (4:4:4, 4:2:2, 4:2:0, grey, restart intervals, optimised tables) and
    Adam7-interlaced PNG files, pixel for pixel what the reference's loader (stb_image, framework/src/image.cpp:13-35)
    decodes them to (fixtures made by oracle/make_golden.py from the reference's own object code)."""
    with np.load(os.path.join(GOLD, "image_cases", "expected.npz")) as z:
        want = z[name]
    got = srt.load_image(os.path.join(GOLD, "image_cases", name))
    assert got.shape == want.shape
    assert_bits_equal(got, want.astype(np.float32) / np.float32(255.0), name)


def test_environment_image_matches_reference_decoder():
    """The reference's own environment map (data/cube_map.jpg, a 4000 x 3000 progressive JPEG): SHA-256 of all decoded
    pixels and a 1/16 sub-grid, recorded from the reference's decoder."""
    path = os.path.join(REF_DATA, "cube_map.jpg")
    if not os.path.exists(path):
        pytest.skip("the reference checkout is not present")
    import hashlib
    with np.load(os.path.join(GOLD, "image_cases", "cube_map.npz")) as z:
        sha, shape, grid = z["sha256"].tobytes(), tuple(z["shape"]), z["grid16"]
    scene = srt.Scene()
    scene.load_environment(REF_DATA)  # Scene::environment = loadEnvironment(dataDir)
    got = srt.load_image(path)
    u8 = np.rint(got * 255.0).astype(np.uint8)
    assert u8.shape == shape and np.array_equal(u8[::16, ::16], grid)
    assert hashlib.sha256(u8.tobytes()).digest() == sha
    assert np.array_equal(u8.astype(np.float32) / np.float32(255.0), got)


def test_image_loader_refuses_what_it_cannot_decode(tmp_path):
    good = open(os.path.join(GOLD, "image_cases", "base420_odd.jpg"), "rb").read()
    cases = {"truncated.jpg": good[:len(good) // 2], "arithmetic.jpg": good.replace(b"\xff\xc0", b"\xff\xc9", 1),
             "twelve_bit.jpg": good.replace(b"\xff\xc0\x00\x11\x08", b"\xff\xc0\x00\x11\x0c", 1), "garbage.png": b"\x89PNG\r\n\x1a\n" + b"\0" * 40,
             "empty.jpg": b""}
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        with pytest.raises(srt.SrtError):
            srt.load_image(str(p))


def test_readers_refuse_files_that_point_outside_themselves(tmp_path):
    """Inputs the reference's libraries answer with undefined behaviour (found by tools/fuzz): a Huffman table with
    more codes than its code length can hold, an image header that promises far more pixels than the file can encode,
    face indices beyond the vertex list.  Here they are refused."""
    good = bytearray(open(os.path.join(GOLD, "image_cases", "base444.jpg"), "rb").read())
    at = good.index(b"\xff\xc4")
    bad_table = bytearray(good)
    bad_table[at + 5] = 255  # 255 codes of length 1
    huge = bytearray(good)
    sof = huge.index(b"\xff\xc0")
    huge[sof + 5:sof + 9] = b"\xff\xff\xff\xff"  # 65535 x 65535
    png = bytearray(open(os.path.join(GOLD, "image_cases", "ilace_rgb.png"), "rb").read())
    png[16:24] = b"\x00\xff\xff\xff\x00\xff\xff\xff"  # IHDR width and height (the CRC is not checked, as in stb_image)
    for name, data in {"bad_table.jpg": bad_table, "huge.jpg": huge, "huge.png": png}.items():
        p = tmp_path / name
        p.write_bytes(bytes(data))
        with pytest.raises(srt.SrtError):
            srt.load_image(str(p))
    for k, face in enumerate(("f 1 2 9", "f -1 -2 -7", "f 1/5 2/1 3/1", "f 1//4 2//1 3//1")):
        p = tmp_path / f"bad{k}.obj"
        p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\n" + face + "\n")
        with pytest.raises(srt.SrtError):
            srt.Scene.from_obj(str(p))


def test_missing_file_raises():
    with pytest.raises(srt.SrtError):
        srt.Scene.from_obj("/nonexistent/x.obj")


def test_camera_constants(camera_golden):
    g = camera_golden
    for i, cfg in enumerate(g["cfg"]):
        cam = srt.make_camera(int(cfg[8]), int(cfg[9]), cfg[0], cfg[1], cfg[2:5], cfg[5:8])
        assert np.allclose(np.array(cam.quat[:]), g[f"quat{i}"], rtol=1e-6, atol=1e-7)
        assert np.allclose([cam.half_w, cam.half_h], g[f"half{i}"], rtol=1e-6)
        assert np.allclose(np.array(cam.origin[:]), g[f"origin{i}"], rtol=1e-6, atol=1e-6)


def test_bmp_writer(tmp_path):
    rgb = np.zeros((2, 3, 3), np.float32)
    rgb[0, 0] = [1.5, 0.5, -1]   # top-left: clamps to 255, 127, 0
    rgb[1, 2] = [0.999, 0.0, 1.0]
    path = tmp_path / "x.bmp"
    srt.write_bmp(str(path), rgb)
    b = path.read_bytes()
    assert b[:2] == b"BM" and int.from_bytes(b[18:22], "little") == 3 and int.from_bytes(b[22:26], "little") == 2
    off = int.from_bytes(b[10:14], "little")
    px = np.frombuffer(b[off:], np.uint8).reshape(2, 3, 4)  # bottom row first, BGRA
    assert px[1, 0].tolist() == [0, 127, 255, 255]
    assert px[0, 2].tolist() == [255, 0, 254, 255]


def test_bitmap_file_matches_reference_writer(tmp_path):
    """Screen::writeBitmapToFile: byte for byte the file the reference's writer (stb_image_write behind the glm
    conversion of src/screen.cpp:96-103) made of the same pixels."""
    g = np.load(os.path.join(GOLD, "screen_bmp.npz"))
    path = str(tmp_path / "out.bmp")
    srt.write_bmp(path, g["rgb"])
    got = np.frombuffer(open(path, "rb").read(), np.uint8)
    assert got.shape == g["bmp"].shape and np.array_equal(got, g["bmp"])


def _bloom_by_hand(img, threshold, box):
    """Screen::applyBloomFilter (reference src/screen.cpp:32-78) spelled out in Python on float32 scalars."""
    H, W = img.shape[:2]
    flat = img.reshape(-1, 3).astype(np.float32).copy()
    table = [[flat[w * H + h].copy() if flat[w * H + h].max() > threshold else np.zeros(3, np.float32) for h in range(H)] for w in range(W)]
    half = box // 2
    for w in range(W):
        for h in range(H):
            if half <= w < W - half and half <= h < H - half:
                acc = np.zeros(3, np.float32)
                for a in range(w - half, w + half):
                    for b in range(h - half, h + half):
                        acc = (acc + table[a][b]).astype(np.float32)
                table[w][h] = (acc / np.float32(box * box)).astype(np.float32)
            flat[w * H + h] = (flat[w * H + h] + table[w][h]).astype(np.float32)
    return flat.reshape(H, W, 3)


@pytest.mark.parametrize("h,w,box", [(6, 9, 3), (9, 6, 3), (8, 8, 5), (5, 7, 1), (10, 4, 4)])
def test_bloom_filter(orc, h, w, box):
    rng = np.random.default_rng(h * 100 + w)
    img = rng.random((h, w, 3)).astype(np.float32)
    img[0, 0] = 0.5  # exactly at the threshold: not bright
    got = srt.bloom(img, 0.5, box)
    assert_bits_equal(got, orc.bloom(img, 0.5, box), "mirror vs oracle")
    assert_bits_equal(got, _bloom_by_hand(img, 0.5, box), "mirror vs the filter spelled out")
    if box > 1:
        assert np.abs(got - img).max() > 0.1


def test_tile_list_covers_the_frame_centre_first():
    """tilesOf: the work list of a frame shared between GPUs - every pixel in exactly one tile, edge tiles clipped, the
    centre tile first (the reference hands rows to its OpenMP threads in order, src/render.cpp:184-187; here the order
    decides which GPU is still busy at the end of the frame)."""
    for (w, h, tw, th) in [(3840, 2160, 480, 270), (800, 600, 480, 270), (64, 48, 24, 16), (7, 5, 8, 8), (100, 70, 37, 30)]:
        t = srt.tiles_of(w, h, tw, th)
        cover = np.zeros((h, w), np.int32)
        for x0, y0, x1, y1 in t:
            assert 0 <= x0 < x1 <= w and 0 <= y0 < y1 <= h and x1 - x0 <= tw and y1 - y0 <= th
            cover[y0:y1, x0:x1] += 1
        assert (cover == 1).all() and len(t) == -(-w // tw) * -(-h // th)
        centre = np.hypot((t[:, 0] + t[:, 2]) / 2 - w / 2, (t[:, 1] + t[:, 3]) / 2 - h / 2)
        assert (np.diff(centre) >= -1e-9).all()
    assert len(srt.tiles_of(0, 10, 8, 8)) == 0


@pytest.mark.parametrize("workers,items", [(1, 17), (2, 64), (8, 200), (5, 3), (4, 0)])
def test_work_list_hands_every_item_out_once(workers, items):
    """runWorkers, the hand-out of renderFrames (one host thread per copy of the scene, one atomic counter): every item
    runs exactly once, the work spreads over the threads, and an item that throws stops the hand-out and surfaces."""
    done, by = srt.run_workers_selftest(workers, items)
    assert (done == 1).all() and by.sum() == items
    if items >= 8 * workers and workers > 1:
        assert (by > 0).sum() >= 2  # items take 100 to 400 microseconds: no thread can have drained the list alone
    if items > 4:
        with pytest.raises(srt.SrtError, match="item 3 failed"):
            srt.run_workers_selftest(workers, items, fail_at=3)


def _child(code):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root)


def test_one_hip_runtime_per_process():
    """PyTorch's ROCm wheels ship a HIP runtime of their own.  Whoever loads libsrt_hip.so first and PyTorch afterwards has
    two runtimes in the process, and the second cannot initialise the device.  The package therefore loads PyTorch first
    (one runtime, whichever order the caller imports in), and the library itself names the problem when it finds two."""
    import torch  # noqa: F401  (skips nothing: the container has it)
    count = ("import re; maps = open('/proc/self/maps').read(); "
             "print('RUNTIMES', len({m for m in re.findall(r'\\S*libamdhip64\\S*', maps)}))")
    # the package first, torch second: one runtime, and torch still answers
    r = _child("import importlib; srt = importlib.import_module('simple-ray-tracer_amd'); srt.host(); import torch; "
               "print('CUDA', torch.cuda.is_available()); " + count)
    assert r.returncode == 0 and "RUNTIMES 1" in r.stdout and "CUDA" in r.stdout, r.stdout + r.stderr
    # the bare library first (what an integrator without the package might do), torch second: the library says what is wrong
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "simple-ray-tracer_amd", "csrc", "libsrt_hip.so")
    r = _child(f"import ctypes; L = ctypes.CDLL({lib!r}); import torch; " + count + "; "
               "L.srt_last_error.restype = ctypes.c_char_p; h = ctypes.c_void_p(); rc = L.srt_device_create(0, ctypes.byref(h)); "
               "print('RC', rc, L.srt_last_error().decode())")
    assert r.returncode == 0, r.stderr
    if "RUNTIMES 2" in r.stdout:
        assert "RC -2" in r.stdout and "two HIP runtimes" in r.stdout and "import torch" in r.stdout, r.stdout
    else:  # a PyTorch that uses the system runtime: nothing to complain about
        assert "RUNTIMES 1" in r.stdout, r.stdout
