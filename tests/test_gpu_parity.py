"""Parity of the HIP path (through the C ABI) with the CPU oracle on the same inputs.

Bar: BVH queries are bit exact (hit flag, t bits, winning mesh/triangle, normal, barycentrics, uv, texel);
colours agree within 1/255 per channel (the only operation that is not correctly rounded on both sides is powf)."""
import importlib
import os

import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu

srt = importlib.import_module("simple-ray-tracer_amd")

RGB_TOL = 1.0 / 255.0


@pytest.fixture(scope="module")
def gpu():
    srt.set_device(0)
    return srt


@pytest.fixture(scope="module")
def pair(orc, scenes, gpu):
    cache = {}

    def get(name, extra_lights=None):
        key = (name, None if extra_lights is None else np.asarray(extra_lights, np.float32).tobytes())
        if key not in cache:
            d = dict(scenes(name))
            if extra_lights is not None:
                d["lights"] = np.ascontiguousarray(np.asarray(extra_lights, np.float32).reshape(-1, orc.LIGHT_STRIDE))
            cache[key] = (orc.Bvh(d, orc.F_ACCEL), srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL))
        return cache[key]

    return get


def _rays_for(orc, d, n, seed):
    """Camera rays, rays aimed at random surface points from random origins (t = FLT_MAX and t = 1), and
    rays that start on the surface (reflection / shadow style)."""
    rng = np.random.default_rng(seed)
    cam = orc.make_camera(96, 64)
    ndc = orc.pixel_ndc(96, 64).reshape(-1, 2)
    rays = [orc.camera_rays(cam, ndc[rng.choice(len(ndc), n // 4, replace=False)])]
    verts = d["verts"][:, :3]
    lo, hi = verts.min(0), verts.max(0)
    tgt = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    org = (rng.standard_normal((n, 3)) * 1.5).astype(np.float32)
    dirs = (tgt - org).astype(np.float32)
    t = np.full((n, 1), np.finfo(np.float32).max, np.float32)
    t[n // 2:] = 1.0
    rays.append(np.concatenate([org, dirs, t], 1))
    # origins inside the scene's bounding box
    org2 = (lo + (hi - lo) * rng.random((n // 2, 3))).astype(np.float32)
    dirs2 = rng.standard_normal((n // 2, 3)).astype(np.float32)
    dirs2[:10, 0] = 0.0
    t2 = np.where(rng.random((n // 2, 1)) < 0.5, np.float32(1.0), np.finfo(np.float32).max).astype(np.float32)
    rays.append(np.concatenate([org2, dirs2, t2], 1))
    return np.ascontiguousarray(np.concatenate(rays).astype(np.float32))


@pytest.mark.parametrize("name", ["single_triangle", "cube", "cube_textured", "cornell", "monkey", "teapot", "custom"])
@pytest.mark.parametrize("policy", ["reference", "cull"])
@pytest.mark.parametrize("attrs", [0, 1])
def test_intersect_bit_exact(orc, scenes, pair, name, policy, attrs):
    ob, gb = pair(name)
    n = 600 if (name == "teapot" and policy == "reference") else 4000
    rays = _rays_for(orc, scenes(name), n, 11)
    feat = orc.F_ACCEL | (orc.F_MISS_CULL if policy == "cull" else 0)
    if attrs:
        feat |= orc.F_NORMAL_INTERP | orc.F_TEXTURE
    want, wc = ob.intersect(rays, feat)
    got, grays, gc = gb.intersect(rays, feat)
    assert np.array_equal(got["hit"], want["hit"])
    assert_bits_equal(got["t"], want["t"], "t")
    assert_bits_equal(grays[:, 6], want["t"], "ray.t")
    assert np.array_equal(got["mesh"], want["mesh"]) and np.array_equal(got["tri"], want["tri"])
    h = want["hit"] == 1
    for f in ("normal", "bary", "uv", "kd"):
        assert_bits_equal(got[f][h], want[f][h], f)
    assert 0 < h.sum() < len(h)
    # same visiting order => same work
    assert gc["inner_visits"] == wc["inner_visits"] and gc["tri_tests"] == wc["tri_tests"]
    assert gc["path_rays"] == len(rays)


@pytest.mark.parametrize("name", ["cube", "monkey"])
def test_intersect_without_accel_structure(orc, scenes, pair, name):
    ob, gb = pair(name)
    rays = _rays_for(orc, scenes(name), 1500, 5)
    want, _ = ob.intersect(rays, 0)
    got, _, gc = gb.intersect(rays, 0)
    assert np.array_equal(got["hit"], want["hit"]) and np.array_equal(got["tri"], want["tri"])
    assert_bits_equal(got["t"], want["t"], "t")
    assert gc["tri_tests"] == len(rays) * len(scenes(name)["tris"])


def _lattice_scene(orc, n=48):
    """Two meshes of a regular grid of quads: thousands of exactly equal centroid coordinates on every axis."""
    d = orc.empty_scene()
    verts, tris = [], []
    for z in (0.0, 0.5):
        g = np.arange(n + 1, dtype=np.float32) / np.float32(n)
        x, y = np.meshgrid(g, g, indexing="ij")
        v = np.zeros(((n + 1) ** 2, 8), np.float32)
        v[:, 0], v[:, 1], v[:, 2], v[:, 5] = x.ravel(), y.ravel(), z, 1.0
        i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        a = (i * (n + 1) + j).ravel().astype(np.uint32)
        t = np.empty((2 * n * n, 3), np.uint32)
        t[0::2] = np.stack([a, a + 1, a + n + 2], 1)
        t[1::2] = np.stack([a, a + n + 2, a + n + 1], 1)
        verts.append(v), tris.append(t)
    d["verts"], d["tris"] = np.concatenate(verts), np.concatenate(tris)
    d["mesh_vert_off"] = np.array([0, len(verts[0]), len(d["verts"])], np.uint32)
    d["mesh_tri_off"] = np.array([0, len(tris[0]), len(d["tris"])], np.uint32)
    d["materials"] = np.array([[1, 1, 1, 0.2, 0.2, 0.2, 10.0, 1.0]] * 2, np.float32)
    d["mesh_tex"] = np.array([-1, -1], np.int32)
    return d


@pytest.mark.parametrize("name", ["single_triangle", "cube", "cube_textured", "cornell", "monkey", "teapot", "custom", "empty", "lattice", "dragon_standin"])
def test_device_built_scene_equals_host_built(orc, scenes, gpu, name):
    """srt_scene_build (triangle records and node boxes computed by kernels, node numbering from the triangle count)
    leaves on the device exactly the arrays the host mirror's builder + flatten(0) produce: every bit of every node
    box, plane, cross product and attribute, the signs of zeros included."""
    if name == "dragon_standin":
        d = importlib.import_module("simple-ray-tracer_amd.standin").dragon_standin()
    else:
        d = orc.empty_scene() if name == "empty" else (_lattice_scene(orc) if name == "lattice" else scenes(name))
    scene = srt.Scene.from_arrays(d)
    got = srt.Bvh(scene, srt.F_ACCEL).download()
    hb = srt.HostBvh(scene, srt.F_ACCEL)
    want = hb.flatten(0)
    assert (got["num_levels"], got["num_leaves"], got["root_ref"]) == (hb.num_levels, hb.num_leaves, want["root_ref"])
    assert got["root_box"].tobytes() == want["root_box"].tobytes()
    for key in ("nodes", "tris", "attrs", "scene_order"):
        assert got[key].shape == want[key].shape, key
        assert got[key].tobytes() == want[key].tobytes(), key


def _soup_scene(orc, n, seed, keys_x=None):
    """n small triangles on a coarse coordinate grid (many exactly equal centroid coordinates on every axis).  With
    keys_x the three corners of triangle i share x = keys_x[i], so its centroid's x is exactly that number."""
    rng = np.random.default_rng(seed)
    d = orc.empty_scene()
    base = rng.integers(0, 24, (n, 1, 3)).astype(np.float32)
    corners = base + rng.integers(0, 4, (n, 3, 3)).astype(np.float32) * np.float32(0.25)
    if keys_x is not None:
        corners[:, :, 0] = np.asarray(keys_x, np.float32)[:, None]
    v = np.zeros((3 * n, 8), np.float32)
    v[:, :3], v[:, 5] = corners.reshape(-1, 3), 1.0
    d["verts"], d["tris"] = v, np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    d["mesh_vert_off"], d["mesh_tri_off"] = np.array([0, 3 * n], np.uint32), np.array([0, n], np.uint32)
    d["materials"] = np.array([[1, 1, 1, 0.2, 0.2, 0.2, 10.0, 1.0]], np.float32)
    d["mesh_tex"] = np.array([-1], np.int32)
    return d


def _assert_device_build_equals_host_build(d):
    scene = srt.Scene.from_arrays(d)
    got = srt.Bvh(scene, srt.F_ACCEL).download()
    hb = srt.HostBvh(scene, srt.F_ACCEL)
    want = hb.flatten(0)
    assert (got["num_levels"], got["num_leaves"]) == (hb.num_levels, hb.num_leaves)
    for key in ("nodes", "tris", "attrs", "scene_order"):
        assert got[key].tobytes() == want[key].tobytes(), key


@pytest.mark.parametrize("n", [2, 3, 16, 17, 18, 33, 100, 1023, 1024, 1025, 1026, 2047, 2048, 2049, 3073, 5000, 70001, 300000])
def test_device_sort_reproduces_std_sort_order(orc, gpu, n, monkeypatch, capfd):
    """The device-side sort leaves equal keys in the order libstdc++'s introsort leaves them, for range sizes on both
    sides of every threshold it has (16: insertion sort only; 1024: the tile kernel's limit; 2048: two tiles)."""
    monkeypatch.setenv("SRT_BUILD_TRACE", "1")
    _assert_device_build_equals_host_build(_soup_scene(orc, n, 100 + n))
    err = capfd.readouterr().err
    assert "device sort" in err and "gave up" not in err and "host sort" not in err


@pytest.mark.parametrize("n", [600, 5000])
def test_device_sort_hands_depth_limit_cases_to_the_host(orc, gpu, n, monkeypatch, capfd):
    """Keys on which std::sort exhausts its depth limit and heapsorts (tests/golden/make_sort_killer.py): the device
    sort does not reproduce heapsort; it reports the case and the build sorts on the host, with the same result."""
    keys = np.load(os.path.join(os.path.dirname(__file__), "golden", "sort_killer.npz"))[f"keys{n}"]
    monkeypatch.setenv("SRT_BUILD_TRACE", "1")
    _assert_device_build_equals_host_build(_soup_scene(orc, n, 7, keys_x=keys))
    err = capfd.readouterr().err
    assert "device sort (gave up)" in err and "host sort" in err


def test_build_with_host_sort(orc, scenes, gpu, monkeypatch, capfd):
    monkeypatch.setenv("SRT_BUILD_SORT", "host")
    monkeypatch.setenv("SRT_BUILD_TRACE", "1")
    _assert_device_build_equals_host_build(scenes("teapot"))
    err = capfd.readouterr().err
    assert "host sort" in err and "device sort" not in err


def test_empty_scene_and_empty_batch(orc, gpu):
    d = orc.empty_scene()
    gb = srt.Bvh(srt.Scene.from_arrays(d))
    assert (gb.num_levels, gb.num_leaves) == (1, 1)
    rays = np.array([[0, 0, -3, 0, 0, 1, 3.4e38]], np.float32)
    hits, _, _ = gb.intersect(rays, srt.F_ACCEL)
    assert hits["hit"][0] == 0 and hits["tri"][0] == -1
    hits, _, _ = gb.intersect(np.zeros((0, 7), np.float32), srt.F_ACCEL)
    assert len(hits) == 0
    img = gb.render(srt.make_camera(8, 8), srt.F_ACCEL | srt.F_SHADING)
    assert not img["rgb"].any()


def test_generate_rays_bit_exact(orc, gpu, camera_golden):
    g = camera_golden
    for i in range(len(g["cfg"])):
        cam = srt.Camera()
        cam.origin[:] = g[f"origin{i}"].tolist()
        cam.quat[:] = g[f"quat{i}"].tolist()
        cam.half_w, cam.half_h = float(g[f"half{i}"][0]), float(g[f"half{i}"][1])
        cam.width, cam.height = int(g["cfg"][i][8]), int(g["cfg"][i][9])
        assert_bits_equal(srt.generate_rays(cam, g[f"ndc{i}"]), g[f"rays{i}"], f"camera rays cfg {i}")


SH = srt.F_ACCEL | srt.F_SHADING
FEATURE_SETS = {
    "albedo": srt.F_ACCEL,
    "shading": SH,
    "hard_shadow": SH | srt.F_HARD_SHADOW,
    "recursive": SH | srt.F_HARD_SHADOW | srt.F_RECURSIVE,
    "interp_texture": SH | srt.F_HARD_SHADOW | srt.F_RECURSIVE | srt.F_NORMAL_INTERP | srt.F_TEXTURE,
    "no_accel": srt.F_SHADING | srt.F_HARD_SHADOW | srt.F_RECURSIVE,
}


POLICY = {"reference": 0, "cull": srt.F_MISS_CULL, "coarse": srt.F_MISS_CULL | srt.F_COARSE_SHADOW}


def _compare_frames(got, want, what):
    assert np.array_equal(got["mesh"], want["mesh"]) and np.array_equal(got["tri"], want["tri"]), what
    assert_bits_equal(got["t"], want["t"], what + " primary t")
    err = np.abs(got["rgb"] - want["rgb"]).max()
    assert err <= RGB_TOL, (what, float(err))
    assert want["rgb"].max() > 0.05, what  # the frame is not trivially black
    return float(err)


@pytest.mark.parametrize("name,res", [("cube_textured", (160, 120)), ("cornell", (128, 128)), ("monkey", (128, 96)), ("teapot", (96, 54))])
@pytest.mark.parametrize("fs", list(FEATURE_SETS))
@pytest.mark.parametrize("policy", ["reference", "cull", "coarse"])
def test_render_matches_oracle(orc, pair, name, res, fs, policy):
    if fs == "no_accel" and name == "teapot":
        pytest.skip("brute force over 15 704 triangles: covered by the smaller scenes")
    if policy == "reference" and name == "teapot" and fs not in ("albedo", "hard_shadow"):
        pytest.skip("full-tree shadow walks on the oracle: two feature sets are enough at this size")
    ob, gb = pair(name)
    feat = FEATURE_SETS[fs] | POLICY[policy]
    want = ob.render(orc.make_camera(*res), feat)
    got = gb.render(srt.make_camera(*res), feat)
    _compare_frames(got, want, f"{name}/{fs}/{policy}")
    wc, gc = want["counters"], got["counters"]
    assert (gc["path_rays"], gc["shadow_rays"], gc["shaded_hits"]) == (wc["path_rays"], wc["shadow_rays"], wc["shaded_hits"])


@pytest.mark.parametrize("nsx,nsy", [(2, 2), (3, 3), (4, 2)])
def test_multisample(orc, pair, nsx, nsy):
    ob, gb = pair("monkey")
    feat = FEATURE_SETS["recursive"] | srt.F_MULTISAMPLE | srt.F_MISS_CULL
    want = ob.render(orc.make_camera(80, 60), feat, nsx=nsx, nsy=nsy)
    got = gb.render(srt.make_camera(80, 60), feat, nsx=nsx, nsy=nsy)
    _compare_frames(got, want, f"multisample {nsx}x{nsy}")
    assert got["counters"]["path_rays"] == want["counters"]["path_rays"] >= 80 * 60 * nsx * nsy


@pytest.mark.parametrize("nsx,nsy", [(2, 2), (4, 2), (4, 4), (3, 2)])
def test_multisample_without_reflections_odd_sizes(orc, pair, nsx, nsy):
    """Frames with a multiple of four samples per pixel and no parked reflection levels are resolved by the path that
    takes the pixels in the order their samples are stored (three 16-byte loads per four samples); 3 x 2 takes the
    pixel-order path.  Sizes that are no multiples of the 8 x 8 sample blocks, whole and as a rectangle inside the frame:
    same sums in the same order as the oracle's, so the colours agree bit for bit where no powf is involved (albedo)."""
    ob, gb = pair("monkey")
    for feat in (srt.F_ACCEL | srt.F_MULTISAMPLE | srt.F_MISS_CULL, SH | srt.F_HARD_SHADOW | srt.F_MULTISAMPLE | srt.F_MISS_CULL | srt.F_COARSE_SHADOW):
        for rect in (None, (5, 3, 31, 26)):
            want = ob.render(orc.make_camera(37, 29), feat, nsx=nsx, nsy=nsy, rect=rect)
            got = gb.render(srt.make_camera(37, 29), feat, nsx=nsx, nsy=nsy, rect=rect)
            _compare_frames(got, want, f"multisample {nsx}x{nsy} 37x29 rect {rect}")
            if not (feat & srt.F_SHADING):
                assert np.array_equal(got["rgb"].view(np.uint32), want["rgb"].view(np.uint32))


@pytest.mark.parametrize("max_depth", [0, 1, 4, 5])
def test_reflection_depth(orc, pair, max_depth):
    ob, gb = pair("monkey")
    feat = FEATURE_SETS["recursive"] | srt.F_MISS_CULL
    want = ob.render(orc.make_camera(64, 48), feat, max_depth=max_depth)
    got = gb.render(srt.make_camera(64, 48), feat, max_depth=max_depth)
    _compare_frames(got, want, f"max_depth {max_depth}")
    assert got["counters"]["path_rays"] == want["counters"]["path_rays"]


@pytest.mark.parametrize("name,extra", [("monkey", 0), ("teapot", srt.F_NORMAL_INTERP), ("cornell", 0), ("cube_textured", srt.F_TEXTURE | srt.F_NORMAL_INTERP)])
def test_reflection_sum_in_reference_order(orc, pair, name, extra):
    """Lo_0 + ks_0 * (Lo_1 + ks_1 * (...)) is evaluated innermost product first, as the reference's recursion does
    (src/render.cpp:109).  Without the shading switch a hit's own colour is its kd - no powf anywhere - so every
    operation is correctly rounded on both sides and the frames must agree bit for bit, multi-bounce pixels included."""
    ob, gb = pair(name)
    feat = srt.F_ACCEL | srt.F_RECURSIVE | srt.F_MISS_CULL | extra
    want = ob.render(orc.make_camera(96, 72), feat, max_depth=5)
    got = gb.render(srt.make_camera(96, 72), feat, max_depth=5)
    assert_bits_equal(got["rgb"], want["rgb"], name + " reflection sums")
    if name in ("monkey", "teapot"):  # concave and specular all over: some pixels see more than one bounce
        assert (ob.render(orc.make_camera(96, 72), feat, max_depth=1)["rgb"] != want["rgb"]).any()
    ms = srt.F_MULTISAMPLE
    assert_bits_equal(gb.render(srt.make_camera(40, 30), feat | ms, nsx=2, nsy=2, max_depth=5)["rgb"],
                      ob.render(orc.make_camera(40, 30), feat | ms, nsx=2, nsy=2, max_depth=5)["rgb"], name + " 2x2")


@pytest.mark.parametrize("name", ["cornell_parallelogram", "cube"])
def test_soft_shadows(orc, pair, name):
    # parallelogram light (10 x 10 samples) and segment light (100 samples); both sides draw their samples
    # from the same counter-based generator, so the frames agree to rounding
    ob, gb = pair(name)
    feat = SH | srt.F_SOFT_SHADOW | srt.F_MISS_CULL
    want = ob.render(orc.make_camera(64, 64), feat, seed=7)
    got = gb.render(srt.make_camera(64, 64), feat, seed=7)
    _compare_frames(got, want, name + " soft shadows")
    assert got["counters"]["shadow_rays"] == want["counters"]["shadow_rays"] == 100 * want["counters"]["shaded_hits"]
    other = gb.render(srt.make_camera(64, 64), feat, seed=8)
    assert np.abs(other["rgb"] - got["rgb"]).max() > 0  # the seed matters


def test_mixed_lights(orc, pair):
    lights = [orc.point_light((0, 0.58, 0)), orc.segment_light((0.25, 0.5, 0), (-0.25, 0.5, 0), (1, 0, 0), (0, 0, 1)),
              orc.parallelogram_light((-0.2, 0.5, 0), (0.4, 0, 0), (0, 0, 0.4), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1))]
    ob, gb = pair("cornell", lights)
    feat = SH | srt.F_HARD_SHADOW | srt.F_SOFT_SHADOW | srt.F_RECURSIVE | srt.F_MISS_CULL
    want = ob.render(orc.make_camera(48, 48), feat)
    got = gb.render(srt.make_camera(48, 48), feat)
    _compare_frames(got, want, "three light types")


def test_many_lights(orc, pair):
    """More lights than a block of the shadow kernel keeps in LDS (16): the later ones are read from global memory.
    18 point lights and, last, a segment light: 18 + 100 shadow queries per hit (a chunk of 64 queries spans hits), and
    with hard shadows only 18 per hit (several hits per chunk)."""
    rng = np.random.default_rng(4)
    lights = [orc.point_light((float(x), 0.55, float(z)), (0.08, 0.08, 0.08)) for x, z in rng.uniform(-0.4, 0.4, (18, 2))]
    lights.append(orc.segment_light((0.25, 0.5, 0.1), (-0.25, 0.5, -0.1), (0.3, 0, 0), (0, 0, 0.3)))
    ob, gb = pair("cornell", lights)
    for feat, what in ((SH | srt.F_HARD_SHADOW | srt.F_SOFT_SHADOW | srt.F_MISS_CULL, "19 lights"),
                       (SH | srt.F_HARD_SHADOW | srt.F_RECURSIVE | srt.F_MISS_CULL, "19 lights, hard shadows only")):
        want = ob.render(orc.make_camera(40, 40), feat, seed=3)
        got = gb.render(srt.make_camera(40, 40), feat, seed=3)
        _compare_frames(got, want, what)
        assert got["counters"]["shadow_rays"] == want["counters"]["shadow_rays"] > 0


@pytest.mark.parametrize("name,res", [("cornell_parallelogram", (96, 96)), ("monkey", (128, 96)), ("teapot", (96, 54)), ("cube", (80, 60))])
def test_coarse_shadow_boxes(orc, pair, name, res):
    """Shadow queries on the four-wide coarse nodes (16-bit outward-rounded boxes of the grandchildren): the frame with
    soft and hard shadows and reflections equals the oracle's under the same policy - node visits and triangle tests
    included, the oracle restates the walk - and equals the frame on full-precision boxes bit for bit: a coarse box can
    only add nodes to a query's walk, the triangle tests decide."""
    ob, gb = pair(name)
    base = SH | srt.F_HARD_SHADOW | srt.F_SOFT_SHADOW | srt.F_RECURSIVE | srt.F_MISS_CULL
    want = ob.render(orc.make_camera(*res), base | srt.F_COARSE_SHADOW)
    got = gb.render(srt.make_camera(*res), base | srt.F_COARSE_SHADOW, flags=srt.RENDER_STATS)
    exact = gb.render(srt.make_camera(*res), base, flags=srt.RENDER_STATS)
    _compare_frames(got, want, f"{name}/coarse shadow boxes")
    assert np.array_equal(got["rgb"].view(np.uint32), exact["rgb"].view(np.uint32))
    gc, ec, wc = got["counters"], exact["counters"], want["counters"]
    assert gc["shadow_rays"] == ec["shadow_rays"] == wc["shadow_rays"] > 0
    assert gc["inner_visits"] == wc["inner_visits"] and gc["tri_tests"] == wc["tri_tests"]
    # a superset of the leaves, by a small margin; two levels per visit: fewer node visits than the level-by-level walk
    # (not in the two tiny scenes, where most shadow rays miss the root box: the wide walk has no root test and pays one
    # visit to find that out)
    assert ec["tri_tests"] <= gc["tri_tests"] <= 1.1 * ec["tri_tests"]
    if name in ("monkey", "teapot"):
        assert gc["inner_visits"] < ec["inner_visits"]


@pytest.mark.parametrize("name,res", [("cornell_parallelogram", (96, 96)), ("monkey", (128, 96)), ("teapot", (96, 54))])
def test_shadow_walk_on_compact_and_full_triangle_records(scenes, gpu, monkeypatch, name, res):
    """The coarse shadow walk tests triangles on 48-byte records (normal + vertices; the plane offset and the second
    operands of the inside tests recomputed).  SRT_FULL_SHADOW_TRIANGLES, read when a scene is created, keeps the walk on
    the 96-byte records: same frame bit for bit, same work."""
    d = scenes(name)
    feat = SH | srt.F_HARD_SHADOW | srt.F_SOFT_SHADOW | srt.F_RECURSIVE | srt.F_MISS_CULL | srt.F_COARSE_SHADOW
    compact = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL).render(srt.make_camera(*res), feat, flags=srt.RENDER_STATS)
    monkeypatch.setenv("SRT_FULL_SHADOW_TRIANGLES", "1")
    gb = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)
    monkeypatch.delenv("SRT_FULL_SHADOW_TRIANGLES")
    full = gb.render(srt.make_camera(*res), feat, flags=srt.RENDER_STATS)
    assert np.array_equal(compact["rgb"].view(np.uint32), full["rgb"].view(np.uint32))
    for key in ("path_rays", "shadow_rays", "shaded_hits", "inner_visits", "tri_tests", "box_tests"):  # (wave steps depend on scheduling)
        assert compact["counters"][key] == full["counters"][key], key
    assert compact["counters"]["tri_tests"] > 0


@pytest.mark.parametrize("cap", [0, 2, 5])
def test_coarse_shadow_stack_overflow(orc, scenes, gpu, monkeypatch, cap):
    """A wide node can leave three siblings pending, the stack holds levels + 1 entries: a query that wants more is
    answered by the exact walk.  With the capacity lowered (SRT_WIDE_CAP, read when a scene is created, by the library
    and by the oracle) that path runs for many queries - all of them at 0: the frames must not change."""
    d = dict(scenes("monkey"))
    feat = SH | srt.F_HARD_SHADOW | srt.F_SOFT_SHADOW | srt.F_MISS_CULL | srt.F_COARSE_SHADOW
    d["lights"] = np.concatenate([np.asarray(d["lights"], np.float32).reshape(-1, orc.LIGHT_STRIDE),
                                  np.asarray(orc.parallelogram_light((-0.3, 1.4, -0.3), (0.6, 0, 0), (0, 0, 0.6), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1)),
                                             np.float32).reshape(1, -1)])
    gb0 = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)
    full = gb0.render(srt.make_camera(96, 72), feat, flags=srt.RENDER_STATS)
    monkeypatch.setenv("SRT_WIDE_CAP", str(cap))
    ob, gb = orc.Bvh(d, orc.F_ACCEL), srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)
    monkeypatch.delenv("SRT_WIDE_CAP")
    want = ob.render(orc.make_camera(96, 72), feat)
    got = gb.render(srt.make_camera(96, 72), feat, flags=srt.RENDER_STATS)
    _compare_frames(got, want, f"wide stack capacity {cap}")
    assert np.array_equal(got["rgb"].view(np.uint32), full["rgb"].view(np.uint32))
    # the lowered capacity did send queries the other way, and both sides count the same work on it (the oracle's
    # fallback stops, like the kernels' exact shadow walk, at the first leaf that leaves t below 0.99)
    gc, wc = got["counters"], want["counters"]
    assert gc["inner_visits"] != full["counters"]["inner_visits"]
    assert gc["shadow_rays"] == wc["shadow_rays"] == full["counters"]["shadow_rays"]
    assert gc["inner_visits"] == wc["inner_visits"] and gc["tri_tests"] == wc["tri_tests"]


@pytest.mark.parametrize("policy", ["reference", "cull", "coarse"])
@pytest.mark.parametrize("name,behind", [("cornell", 0.004), ("monkey", 0.0005)])
def test_shadow_hit_just_before_light(orc, pair, name, behind, policy):
    # Lights just behind a surface: seen from far enough away the surface is hit at 0.99 <= t <= 1, which the
    # reference does not count as occlusion (light.cpp:60) but walks on from with the smaller t.
    cull = srt.F_MISS_CULL if policy != "reference" else 0
    ob, _ = pair(name)
    cam = orc.make_camera(64, 48)
    rays = orc.camera_rays(cam, orc.pixel_ndc(64, 48).reshape(-1, 2))
    prim, _ = ob.intersect(rays, orc.F_ACCEL | cull)
    hit = np.nonzero(prim["hit"])[0]
    pts = rays[hit, :3] + rays[hit, 3:6] * prim["t"][hit, None]
    lights = []
    for i in hit[:: max(1, len(hit) // 3)][:3]:
        q = rays[i, :3] + rays[i, 3:6] * prim["t"][i]
        n = prim["normal"][i] / np.linalg.norm(prim["normal"][i])
        if np.dot(n, rays[i, 3:6]) < 0:
            n = -n  # away from the camera: behind the surface
        lights.append(orc.point_light(tuple((q + behind * n).astype(np.float32))))
    ob, gb = pair(name, lights)
    # the case is present: shadow rays of the primary hits that end with a hit in [0.99, 1]
    near = 0
    for lt in lights:
        lp = np.asarray(lt, np.float32).ravel()[1:4]
        dv = (lp[None, :] - pts).astype(np.float32)
        org = pts + dv / np.linalg.norm(dv, axis=1, keepdims=True) * np.float32(0.001)
        sr = np.concatenate([org, dv, np.ones((len(pts), 1), np.float32)], axis=1).astype(np.float32)
        want, _ = ob.intersect(sr, orc.F_ACCEL | cull)
        got, _, _ = gb.intersect(sr, srt.F_ACCEL | cull)
        assert np.array_equal(got["hit"], want["hit"]) and np.array_equal(got["tri"], want["tri"])
        assert_bits_equal(got["t"], want["t"], "shadow-ray t")
        near += int(((want["hit"] != 0) & (want["t"] >= 0.99)).sum())
    assert near > 30, near
    feat = SH | srt.F_HARD_SHADOW | srt.F_RECURSIVE | POLICY[policy]
    want = ob.render(orc.make_camera(64, 48), feat)
    got = gb.render(srt.make_camera(64, 48), feat)
    _compare_frames(got, want, f"lights just behind surfaces, {name}/{policy}")


def test_final_color_batch(orc, scenes, pair):
    ob, gb = pair("cornell")
    rays = _rays_for(orc, scenes("cornell"), 2000, 3)
    feat = FEATURE_SETS["recursive"] | srt.F_MISS_CULL
    want, _ = ob.final_color(rays, feat)
    got = gb.final_color(rays, feat)
    assert np.abs(got - want).max() <= RGB_TOL and want.max() > 0.05


def test_tiles_equal_whole_frame(gpu, pair):
    _, gb = pair("monkey")
    cam = srt.make_camera(100, 70)
    feat = FEATURE_SETS["recursive"] | srt.F_MULTISAMPLE | srt.F_MISS_CULL
    whole = gb.render(cam, feat, nsx=2, nsy=2)
    again = gb.render(cam, feat, nsx=2, nsy=2)
    assert np.array_equal(whole["rgb"].view(np.uint32), again["rgb"].view(np.uint32))  # reproducible
    acc = np.zeros_like(whole["rgb"])
    for rect in [(0, 0, 37, 30), (37, 0, 100, 30), (0, 30, 64, 70), (64, 30, 100, 70)]:
        part = gb.render(cam, feat, nsx=2, nsy=2, rect=rect)["rgb"]
        x0, y0, x1, y1 = rect
        acc[70 - y1:70 - y0, x0:x1] = part[70 - y1:70 - y0, x0:x1]
        outside = part.copy()
        outside[70 - y1:70 - y0, x0:x1] = 0
        assert not outside.any()  # nothing written outside the rectangle
    assert np.array_equal(acc.view(np.uint32), whole["rgb"].view(np.uint32))


def test_device_output_two_contexts_two_streams(gpu, pair, scenes):
    """What bench.py does with a tiled frame: tiles rendered straight into device memory (tile-local buffers), by
    two back-end contexts on two streams at once; assembled they are the host-output frame bit for bit."""
    import torch
    _, gb = pair("monkey")
    srt.set_device(0, context=1)
    gb2 = srt.Bvh(srt.Scene.from_arrays(dict(scenes("monkey"))), srt.F_ACCEL)
    srt.set_device(0, context=0)
    W, H, tw, th = 96, 64, 32, 32
    cam = srt.make_camera(W, H)
    feat = FEATURE_SETS["recursive"] | srt.F_MULTISAMPLE | srt.F_MISS_CULL
    whole = gb.render(cam, feat, nsx=2, nsy=2, primary=False)["rgb"]
    tiling = importlib.import_module("simple-ray-tracer_amd.distributed")
    rects = tiling.tile_rects(W, H, tw, th)
    bufs = torch.zeros((len(rects), th, tw, 3), dtype=torch.float32, device="cuda")
    lanes = [(gb, torch.cuda.Stream()), (gb2, torch.cuda.Stream())]
    torch.cuda.synchronize()
    for k, rect in enumerate(rects):
        b, stream = lanes[k % 2]
        p = b.params(cam, feat, 2, 2, 5, 0, rect, 0)
        b.render_device(p, bufs[k].data_ptr(), tw, th, rect[0], rect[1], stream.cuda_stream)
    torch.cuda.synchronize()
    frame = np.zeros((H, W, 3), np.float32)
    tiles = bufs.cpu().numpy()
    for k, (x0, y0, x1, y1) in enumerate(rects):
        frame[H - y1:H - y0, x0:x1] = tiles[k, th - (y1 - y0):, :x1 - x0]
    assert np.array_equal(frame.view(np.uint32), whole.view(np.uint32))
    c1, c2 = gb.read_counters(reset=True), gb2.read_counters(reset=True)
    assert c2["path_rays"] > 0 and c1["path_rays"] > 0  # both contexts did render


def test_argument_errors(gpu, pair):
    _, gb = pair("cube")
    cam = srt.make_camera(16, 16)
    with pytest.raises(srt.SrtError):
        gb.render(cam, srt.F_ACCEL, rect=(0, 0, 17, 16))
    with pytest.raises(srt.SrtError):
        gb.render(cam, srt.F_ACCEL, rect=(4, 4, 4, 8))
    with pytest.raises(srt.SrtError):
        gb.render(cam, srt.F_ACCEL | srt.F_MULTISAMPLE, nsx=0, nsy=2)
    with pytest.raises(srt.SrtError):
        gb.render(cam, srt.F_ACCEL, max_depth=99)


def test_teapot_full_size_properties(gpu, pair):
    """1920 x 1080 teapot frame (primary + shadow rays): too large for the oracle in test time, so checked
    through size-independent properties: run-to-run identity, tile decomposition, the recorded hit fraction, and
    agreement between the reference visiting rule and the culling one."""
    _, gb = pair("teapot")
    cam = srt.make_camera(1920, 1080)
    feat = SH | srt.F_HARD_SHADOW
    cull = gb.render(cam, feat | srt.F_MISS_CULL)
    ref = gb.render(cam, feat)
    n = 1920 * 1080
    assert cull["counters"]["path_rays"] == n
    same = (cull["tri"] == ref["tri"]).sum()
    assert same >= 0.9999 * n
    assert (np.abs(cull["rgb"] - ref["rgb"]).max(axis=2) <= RGB_TOL).mean() >= 0.9999
    hits = cull["counters"]["shaded_hits"]
    assert 0.06 * n < hits < 0.08 * n  # 6.85 % at 480 x 270 in the recorded reference run
    assert cull["counters"]["shadow_rays"] == hits
    half = gb.render(cam, feat | srt.F_MISS_CULL, rect=(0, 0, 1920, 540))["rgb"]
    assert np.array_equal(half[540:].view(np.uint32), cull["rgb"][540:].view(np.uint32))


@pytest.fixture(scope="module")
def dragon_pair(orc, gpu):
    """The bench's procedural dragon stand-in: more than 2^17 triangles, so the builder hits its depth cap and
    leaves hold two triangles each."""
    standin = importlib.import_module("simple-ray-tracer_amd.standin")
    d = standin.dragon_standin()
    d["lights"] = standin.dragon_lights()
    ob, gb = orc.Bvh(d, orc.F_ACCEL), srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)
    assert gb.num_levels == ob.num_levels == 18 and gb.num_leaves == ob.num_leaves == 131072
    return d, ob, gb


def test_dragon_standin_queries(orc, dragon_pair):
    d, ob, gb = dragon_pair
    rays = _rays_for(orc, d, 6000, 21)
    for feat in (orc.F_ACCEL | orc.F_MISS_CULL | orc.F_NORMAL_INTERP, orc.F_ACCEL | orc.F_MISS_CULL):
        want, wc = ob.intersect(rays, feat)
        got, _, gc = gb.intersect(rays, feat)
        assert np.array_equal(got["hit"], want["hit"]) and np.array_equal(got["tri"], want["tri"])
        assert_bits_equal(got["t"], want["t"], "t")
        assert_bits_equal(got["normal"][want["hit"] == 1], want["normal"][want["hit"] == 1], "normal")
        assert gc["inner_visits"] == wc["inner_visits"] and gc["tri_tests"] == wc["tri_tests"]
    # the reference visiting rule on a subset (each unoccluded t = 1 ray walks all 262 143 nodes on the CPU)
    sub = rays[::40]
    want, wc = ob.intersect(sub, orc.F_ACCEL)
    got, _, gc = gb.intersect(sub, orc.F_ACCEL)
    assert np.array_equal(got["tri"], want["tri"]) and gc["inner_visits"] == wc["inner_visits"]
    assert_bits_equal(got["t"], want["t"], "t (reference rule)")


def test_dragon_standin_frame(orc, dragon_pair):
    """The bench workload's feature set (8 spp as a 4 x 2 grid, point light with hard shadows, parallelogram light
    with soft shadows) at a resolution the oracle renders in seconds."""
    d, ob, gb = dragon_pair
    feat = SH | srt.F_HARD_SHADOW | srt.F_SOFT_SHADOW | srt.F_MULTISAMPLE | srt.F_MISS_CULL
    want = ob.render(orc.make_camera(96, 54), feat, nsx=4, nsy=2)
    got = gb.render(srt.make_camera(96, 54), feat, nsx=4, nsy=2)
    _compare_frames(got, want, "dragon stand-in, bench feature set")
    wc, gc = want["counters"], got["counters"]
    assert (gc["path_rays"], gc["shadow_rays"], gc["shaded_hits"]) == (wc["path_rays"], wc["shadow_rays"], wc["shaded_hits"])
    assert gc["shadow_rays"] == 101 * gc["shaded_hits"]
    # with reflections on top (ks = 0.2 everywhere: every hit bounces until the depth cap)
    want = ob.render(orc.make_camera(48, 27), feat | srt.F_RECURSIVE, nsx=4, nsy=2, max_depth=4)
    got = gb.render(srt.make_camera(48, 27), feat | srt.F_RECURSIVE, nsx=4, nsy=2, max_depth=4)
    _compare_frames(got, want, "dragon stand-in with 4 bounces")
    assert got["counters"]["path_rays"] == want["counters"]["path_rays"]


def test_dragon_full_size_properties(dragon_pair):
    """The 3840 x 2160 x 8 spp bench frame through size-independent properties: identical when rendered twice,
    identical when cut into tiles, and the shadow-query count follows from the hit count."""
    _, _, gb = dragon_pair
    cam = srt.make_camera(3840, 2160)
    feat = SH | srt.F_HARD_SHADOW | srt.F_SOFT_SHADOW | srt.F_MULTISAMPLE | srt.F_MISS_CULL
    a = gb.render(cam, feat, nsx=4, nsy=2, primary=False)
    assert a["counters"]["path_rays"] == 3840 * 2160 * 8
    assert a["counters"]["shadow_rays"] == 101 * a["counters"]["shaded_hits"] > 0
    strip = gb.render(cam, feat, nsx=4, nsy=2, rect=(0, 810, 3840, 1350), primary=False)["rgb"]
    assert np.array_equal(strip[810:1350].view(np.uint32), a["rgb"][810:1350].view(np.uint32))
    assert a["rgb"][810:1350].max() > 0.05 and np.isfinite(a["rgb"]).all()
    # shadow queries on the coarse four-wide nodes: the same frame bit for bit, whole (1024 queries per draw from the chunk
    # counter) and as a small rectangle (shorter chunks)
    c = gb.render(cam, feat | srt.F_COARSE_SHADOW, nsx=4, nsy=2, primary=False)
    assert c["counters"]["shadow_rays"] == a["counters"]["shadow_rays"]
    assert np.array_equal(c["rgb"].view(np.uint32), a["rgb"].view(np.uint32))
    tile = gb.render(cam, feat | srt.F_COARSE_SHADOW, nsx=4, nsy=2, rect=(1680, 945, 2160, 1215), primary=False)["rgb"]
    assert np.array_equal(tile[945:1215, 1680:2160].view(np.uint32), a["rgb"][945:1215, 1680:2160].view(np.uint32))


@pytest.mark.parametrize("name,max_depth,policy", [("cube_textured", 5, "cull"), ("cube_textured", 4, "reference"), ("monkey", 5, "cull"),
                                                   ("monkey", 4, "cull"), ("monkey", 5, "reference")])
def test_textured_recursive_full_size(orc, pair, name, max_depth, policy):
    """Textured cube and monkey at 1920 x 1080 with normal interpolation, texture lookup and mirror recursion
    (depth cap 5 as the reference has it, and 4): whole frames against the oracle."""
    ob, gb = pair(name)
    feat = FEATURE_SETS["interp_texture"] | (srt.F_MISS_CULL if policy == "cull" else 0)
    want = ob.render(orc.make_camera(1920, 1080), feat, max_depth=max_depth)
    got = gb.render(srt.make_camera(1920, 1080), feat, max_depth=max_depth)
    _compare_frames(got, want, f"{name} 1920x1080 depth {max_depth} {policy}")
    wc, gc = want["counters"], got["counters"]
    assert (gc["path_rays"], gc["shadow_rays"], gc["shaded_hits"]) == (wc["path_rays"], wc["shadow_rays"], wc["shaded_hits"])


@pytest.mark.parametrize("max_depth", [1, 5])
def test_cornell_1024(orc, pair, max_depth):
    """Cornell box 1024 x 1024, hard shadows and mirror reflection (one bounce, and the reference's cap of 5), under
    the reference's visiting rule; at depth 5 the ray total is the one recorded from the reference run."""
    ob, gb = pair("cornell")
    feat = FEATURE_SETS["recursive"]
    want = ob.render(orc.make_camera(1024, 1024), feat, max_depth=max_depth)
    got = gb.render(srt.make_camera(1024, 1024), feat, max_depth=max_depth)
    _compare_frames(got, want, f"cornell 1024 depth {max_depth}")
    gc = got["counters"]
    assert gc["path_rays"] == want["counters"]["path_rays"] and gc["shadow_rays"] == want["counters"]["shadow_rays"]
    if max_depth == 5:
        assert gc["path_rays"] + gc["shadow_rays"] == 1328237


def test_dragon_8k_64spp_window(orc, dragon_pair):
    """A window of the 7680 x 4320, 64 spp (8 x 8 grid) stress frame: sample numbering and the sampler keys of a
    2.1-G-sample frame, against the oracle on the same window."""
    d, ob, gb = dragon_pair
    feat = SH | srt.F_HARD_SHADOW | srt.F_SOFT_SHADOW | srt.F_MULTISAMPLE | srt.F_MISS_CULL
    rect = (4004, 3690, 4052, 3714)  # across the upper silhouette; ragged against the 8 x 8-pixel sample blocks
    want = ob.render(orc.make_camera(7680, 4320), feat, nsx=8, nsy=8, rect=rect)
    got = gb.render(srt.make_camera(7680, 4320), feat, nsx=8, nsy=8, rect=rect)
    ys, xs = slice(4320 - rect[3], 4320 - rect[1]), slice(rect[0], rect[2])
    sub = lambda r: {k: (v[ys, xs] if isinstance(v, np.ndarray) else v) for k, v in r.items()}
    _compare_frames(sub(got), sub(want), "8K 64 spp window")
    assert got["counters"]["path_rays"] == want["counters"]["path_rays"] == 48 * 24 * 64
    assert got["counters"]["shadow_rays"] == want["counters"]["shadow_rays"] > 0


def test_dragon_8k_64spp_window_with_reflections(orc, dragon_pair):
    """The same window of the stress frame as its bench workload renders it: mirror recursion to the reference's
    depth cap on top (ks = 0.2 on the whole model, so every hit sends a reflection ray into the 18-level tree)."""
    d, ob, gb = dragon_pair
    feat = SH | srt.F_HARD_SHADOW | srt.F_SOFT_SHADOW | srt.F_MULTISAMPLE | srt.F_RECURSIVE | srt.F_MISS_CULL
    rect = (4004, 3690, 4052, 3714)
    want = ob.render(orc.make_camera(7680, 4320), feat, nsx=8, nsy=8, rect=rect, max_depth=5)
    got = gb.render(srt.make_camera(7680, 4320), feat, nsx=8, nsy=8, rect=rect, max_depth=5)
    ys, xs = slice(4320 - rect[3], 4320 - rect[1]), slice(rect[0], rect[2])
    sub = lambda r: {k: (v[ys, xs] if isinstance(v, np.ndarray) else v) for k, v in r.items()}
    _compare_frames(sub(got), sub(want), "8K 64 spp window with reflections")
    wc, gc = want["counters"], got["counters"]
    assert (gc["path_rays"], gc["shadow_rays"], gc["shaded_hits"]) == (wc["path_rays"], wc["shadow_rays"], wc["shaded_hits"])
    assert gc["path_rays"] > 48 * 24 * 64 + 1000  # reflection rays were traced


def test_dragon_8k_full_frame_properties(dragon_pair):
    """The whole 7680 x 4320, 64 spp stress frame with reflections (2.12 G camera rays), rendered the way the bench
    renders it: 960 x 540 rectangles straight into one frame buffer in HBM.  Size-independent checks: ray totals
    follow from the hit count, a strip cut differently is bit-identical, pixels are finite and not black."""
    import torch
    _, _, gb = dragon_pair
    W, H = 7680, 4320
    cam = srt.make_camera(W, H)
    feat = SH | srt.F_HARD_SHADOW | srt.F_SOFT_SHADOW | srt.F_MULTISAMPLE | srt.F_RECURSIVE | srt.F_MISS_CULL
    tiling = importlib.import_module("simple-ray-tracer_amd.distributed")
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    gb.read_counters(reset=True)
    for rect in tiling.tile_rects(W, H, 960, 540):
        gb.render_device(gb.params(cam, feat, 8, 8, 5, 0, rect, 0), frame.data_ptr(), W, H, 0, 0, stream)
    torch.cuda.synchronize()
    c = gb.read_counters(reset=True)
    n = W * H * 64
    assert n < c["path_rays"] <= n + c["shaded_hits"]  # every hit below the depth cap reflects once
    assert c["shadow_rays"] == 101 * c["shaded_hits"] and c["shaded_hits"] > 0.1 * n
    strip = torch.zeros((64, W, 3), dtype=torch.float32, device="cuda")
    y0 = 2130  # a strip across all eight columns of rectangles and across the boundary between two of their rows (y = 2160)
    gb.render_device(gb.params(cam, feat, 8, 8, 5, 0, (0, y0, W, y0 + 64), 0), strip.data_ptr(), W, 64, 0, y0, stream)
    torch.cuda.synchronize()
    rows = frame[H - (y0 + 64):H - y0]
    assert torch.equal(rows.view(torch.int32), strip.view(torch.int32))
    assert bool(torch.isfinite(frame).all()) and float(rows.max()) > 0.05 and float(frame.mean()) > 0.005


@pytest.mark.parametrize("name", ["monkey", "teapot"])
def test_sah_binned_tree(orc, scenes, gpu, name):
    """extra.enableBvhSahBinning (reference src/bounding_volume_hierarchy.cpp:96-178): a different tree, same
    traversal; queries and a frame against the oracle built the same way."""
    d = scenes(name)
    ob, gb = orc.Bvh(d, orc.F_ACCEL | orc.F_SAH), srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL | srt.F_SAH)
    assert (gb.num_levels, gb.num_leaves) == (ob.num_levels, ob.num_leaves)
    plain = orc.Bvh(d, orc.F_ACCEL)
    assert (ob.num_levels, ob.num_nodes) != (plain.num_levels, plain.num_nodes) or name == "monkey"
    rays = _rays_for(orc, d, 3000, 9)
    for feat in (orc.F_ACCEL, orc.F_ACCEL | orc.F_MISS_CULL):
        want, _ = ob.intersect(rays, feat)
        got, _, _ = gb.intersect(rays, feat)
        assert np.array_equal(got["hit"], want["hit"]) and np.array_equal(got["tri"], want["tri"])
        assert_bits_equal(got["t"], want["t"], "t")
    for feat in (FEATURE_SETS["recursive"] | srt.F_MISS_CULL, FEATURE_SETS["recursive"] | srt.F_SOFT_SHADOW | srt.F_MISS_CULL | srt.F_COARSE_SHADOW):
        _compare_frames(gb.render(srt.make_camera(96, 64), feat), ob.render(orc.make_camera(96, 64), feat), f"{name} SAH tree")


def _see_through_cornell(orc, scenes):
    """Cornell box (mirror included) with materials made partly transparent: reflection and transmission together,
    so a sample's rays form a binary tree."""
    d = dict(scenes("cornell"))
    m = d["materials"].copy()
    m[:, 7] = np.resize(np.array([0.3, 1.0, 0.7, 0.55], np.float32), len(m))
    d["materials"] = m
    return orc.Bvh(d, orc.F_ACCEL), srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)


@pytest.mark.parametrize("policy", ["reference", "cull"])
@pytest.mark.parametrize("max_depth", [5, 2, 0])
def test_transparency_with_reflection(orc, scenes, gpu, max_depth, policy):
    ob, gb = _see_through_cornell(orc, scenes)
    feat = FEATURE_SETS["recursive"] | srt.F_TRANSPARENCY | (srt.F_MISS_CULL if policy == "cull" else 0)
    want = ob.render(orc.make_camera(96, 72), feat, max_depth=max_depth)
    got = gb.render(srt.make_camera(96, 72), feat, max_depth=max_depth)
    assert np.array_equal(got["tri"], want["tri"])
    assert np.abs(got["rgb"] - want["rgb"]).max() <= RGB_TOL
    wc, gc = want["counters"], got["counters"]
    assert (gc["path_rays"], gc["shadow_rays"], gc["shaded_hits"]) == (wc["path_rays"], wc["shadow_rays"], wc["shaded_hits"])
    if max_depth == 0:
        assert not got["rgb"].any()  # every hit sits at the depth cap: black, as in the reference
    else:
        assert want["rgb"].max() > 0.05 and gc["path_rays"] > 96 * 72 + 1000


def test_transparency_cube_scene(orc, pair):
    # the prebuilt cube is the one reference scene whose materials are transparent (d 0.452632); its light is a
    # segment light, hence soft shadows
    ob, gb = pair("cube")
    for extra in (0, srt.F_RECURSIVE):
        feat = SH | srt.F_SOFT_SHADOW | srt.F_TRANSPARENCY | srt.F_MISS_CULL | extra
        want = ob.render(orc.make_camera(64, 48), feat, seed=3)
        got = gb.render(srt.make_camera(64, 48), feat, seed=3)
        _compare_frames(got, want, "transparent cube")
        assert got["counters"]["path_rays"] == want["counters"]["path_rays"] > 64 * 48
    # the C++ mirror: Features::extra.enableTransparency
    rgb = srt.mirror_render(gb, SH | srt.F_SOFT_SHADOW | srt.F_TRANSPARENCY | srt.F_MISS_CULL, 64, 48)
    want = ob.render(orc.make_camera(64, 48), SH | srt.F_SOFT_SHADOW | srt.F_TRANSPARENCY | srt.F_MISS_CULL)
    assert np.abs(rgb - want["rgb"]).max() <= RGB_TOL
    # plain opaque rendering of the same scene differs: the flag does something
    opaque = gb.render(srt.make_camera(64, 48), SH | srt.F_SOFT_SHADOW | srt.F_MISS_CULL)
    assert np.abs(opaque["rgb"] - want["rgb"]).max() > 0.05


def test_transparency_final_color_batch(orc, scenes, gpu):
    ob, gb = _see_through_cornell(orc, scenes)
    rays = _rays_for(orc, scenes("cornell"), 2000, 5)
    feat = FEATURE_SETS["recursive"] | srt.F_TRANSPARENCY | srt.F_MISS_CULL
    want, _ = ob.final_color(rays, feat)
    got = gb.final_color(rays, feat)
    assert np.abs(got - want).max() <= RGB_TOL and want.max() > 0.05


def _environment_image(w=128, h=96):
    """A cross-shaped cube map stand-in: smooth gradients plus a checker, quantised like a decoded image (u8 / 255)."""
    y, x = np.mgrid[0:h, 0:w]
    r = (x * 255 // (w - 1)).astype(np.uint8)
    g = (y * 255 // (h - 1)).astype(np.uint8)
    b = (((x // 8 + y // 8) % 2) * 200 + 40).astype(np.uint8)
    return np.ascontiguousarray(np.stack([r, g, b], -1).astype(np.float32) / np.float32(255.0))


@pytest.mark.parametrize("policy", ["reference", "cull"])
def test_environment_map(orc, scenes, gpu, policy):
    """extra.enableEnvironmentMapping (reference src/render.cpp:40-81,141-147): camera rays, reflected rays and rays
    continued through transparent surfaces that hit nothing pick up the environment image."""
    env = _environment_image()
    cull = srt.F_MISS_CULL if policy == "cull" else 0
    for name, extra, res in (("monkey", srt.F_RECURSIVE | srt.F_NORMAL_INTERP, (96, 72)), ("cube", srt.F_TRANSPARENCY | srt.F_RECURSIVE, (64, 48)),
                             ("cornell", srt.F_RECURSIVE, (64, 64))):
        d = dict(scenes(name))
        d["lights"] = np.ascontiguousarray(np.stack([orc.point_light((-1, 1.5, -1))]))
        scene = srt.Scene.from_arrays(d)
        scene.set_environment(env)
        ob, gb = orc.Bvh(d, orc.F_ACCEL), srt.Bvh(scene, srt.F_ACCEL)
        feat = SH | srt.F_HARD_SHADOW | extra | srt.F_ENVIRONMENT | cull
        want = ob.render(orc.make_camera(*res), feat, environment=env)
        got = gb.render(srt.make_camera(*res), feat)
        _compare_frames(got, want, f"environment map, {name}/{policy}")
        plain = gb.render(srt.make_camera(*res), feat & ~srt.F_ENVIRONMENT)
        miss = got["tri"] < 0
        assert miss.any() and (got["rgb"][miss].max(axis=1) > 0).mean() > 0.9 and not plain["rgb"][miss].any()
        if name != "cornell":  # the box is closed towards everything but the camera: its mirror sees walls only
            assert np.abs(got["rgb"][~miss] - plain["rgb"][~miss]).max() > 0.02  # reflections / transmission see it too
        rays = _rays_for(orc, d, 800, 17)
        want_rgb, _ = ob.final_color(rays, feat, environment=env)
        assert np.abs(gb.final_color(rays, feat) - want_rgb).max() <= RGB_TOL
    # a scene uploaded without an environment image refuses the flag
    _, bare = orc.Bvh(scenes("cube"), orc.F_ACCEL), srt.Bvh(srt.Scene.from_arrays(scenes("cube")), srt.F_ACCEL)
    with pytest.raises(srt.SrtError, match="environment"):
        bare.render(srt.make_camera(16, 16), SH | srt.F_ENVIRONMENT)


@pytest.mark.parametrize("mode", ["motion_blur", "depth_of_field"])
def test_lens_effects(orc, scenes, pair, mode):
    """extra.enableMotionBlur / enableDepthOfField (reference src/render.cpp:152-178,218-223) with the jitter drawn
    from the shared counter-based generator (the reference uses rand(): parity against the reference itself is
    unpinned for these two, as for soft shadows)."""
    ob, gb = pair("monkey")
    flag = srt.F_MOTION_BLUR if mode == "motion_blur" else srt.F_DEPTH_OF_FIELD
    lens = dict(samples=6, time0=0.1, time1=0.9, direction=(1.0, 0.5, 0.0), focal_length=3, aperture=0.05)
    feat = FEATURE_SETS["recursive"] | srt.F_MISS_CULL | flag
    want = ob.render(orc.make_camera(80, 60), feat, seed=5, lens=lens)
    got = gb.render(srt.make_camera(80, 60), feat, seed=5, lens=lens)
    _compare_frames(got, want, mode)
    wc, gc = want["counters"], got["counters"]
    assert (gc["path_rays"], gc["shadow_rays"], gc["shaded_hits"]) == (wc["path_rays"], wc["shadow_rays"], wc["shaded_hits"])
    assert gc["path_rays"] >= 80 * 60 * 6
    sharp = gb.render(srt.make_camera(80, 60), feat & ~flag)
    assert np.abs(sharp["rgb"] - got["rgb"]).max() > 0.05  # the effect is visible
    assert np.abs(gb.render(srt.make_camera(80, 60), feat, seed=6, lens=lens)["rgb"] - got["rgb"]).max() > 0
    # a rectangle of the frame equals the frame (sampler keys are frame global)
    part = gb.render(srt.make_camera(80, 60), feat, seed=5, lens=lens, rect=(16, 8, 61, 43))["rgb"]
    assert np.array_equal(part[60 - 43:60 - 8, 16:61].view(np.uint32), got["rgb"][60 - 43:60 - 8, 16:61].view(np.uint32))
    # zero samples: the reference's loop does not run, the frame is black
    none = gb.render(srt.make_camera(80, 60), feat, lens=dict(lens, samples=0))
    assert not none["rgb"].any() and not ob.render(orc.make_camera(80, 60), feat, lens=dict(lens, samples=0))["rgb"].any()
    # precedence (src/render.cpp:196-227): multisampling over motion blur over depth of field
    both = feat | srt.F_MOTION_BLUR | srt.F_DEPTH_OF_FIELD
    mb = gb.render(srt.make_camera(80, 60), (feat & ~flag) | srt.F_MOTION_BLUR, seed=5, lens=lens)
    assert np.array_equal(gb.render(srt.make_camera(80, 60), both, seed=5, lens=lens)["rgb"], mb["rgb"])
    ms = gb.render(srt.make_camera(80, 60), (feat & ~flag) | srt.F_MULTISAMPLE, nsx=2, nsy=2)
    assert np.array_equal(gb.render(srt.make_camera(80, 60), both | srt.F_MULTISAMPLE, nsx=2, nsy=2, seed=5, lens=lens)["rgb"], ms["rgb"])
    # the C++ mirror takes the variables from the Scene, as the reference does
    gb.scene.set_lens(mb_samples=6, time0=0.1, time1=0.9, direction=(1.0, 0.5, 0.0), focal_length=3, aperture=0.05, dof_samples=6)
    mirror = srt.mirror_render(gb, feat, 80, 60)
    want0 = ob.render(orc.make_camera(80, 60), feat, seed=0, lens=lens)["rgb"]
    assert np.abs(mirror - want0).max() <= RGB_TOL
    gb.scene.set_lens()


def test_bloom_through_the_mirror(orc, pair):
    """renderRayTracing with extra.enableBloomEffect and the parameters the reference's callers pass (threshold 0.5,
    boxSize 1 -> a 3-box): the frame, then Screen::applyBloomFilter on it."""
    ob, gb = pair("cornell")
    feat = FEATURE_SETS["recursive"] | srt.F_MISS_CULL
    plain = ob.render(orc.make_camera(80, 60), feat)["rgb"]
    want = orc.bloom(plain, 0.5, 3)
    got = srt.mirror_render(gb, feat | srt.F_BLOOM, 80, 60)
    assert np.abs(got - want).max() <= RGB_TOL * 2 and np.abs(want - plain).max() > 0.1


def test_fast_division_is_exact(gpu):
    assert srt.selftest_division(1 << 26, seed=3) == 0


def test_host_mirror_call_sites(orc, scenes, pair):
    """The C++ mirror used the way the reference's own call sites use it."""
    ob, gb = pair("cube_textured")
    feat = FEATURE_SETS["interp_texture"] | srt.F_MISS_CULL
    # renderRayTracing(scene, camera, bvh, screen, features, threshold, boxSize, numRays)
    want = ob.render(orc.make_camera(64, 48), feat)
    got = srt.mirror_render(gb, feat, 64, 48)
    assert np.abs(got - want["rgb"]).max() <= RGB_TOL and want["rgb"].max() > 0.05
    ms = feat | srt.F_MULTISAMPLE
    want = ob.render(orc.make_camera(40, 30), ms, nsx=3, nsy=3)
    got = srt.mirror_render(gb, ms, 40, 30, num_rays=3)
    assert np.abs(got - want["rgb"]).max() <= RGB_TOL
    # Trackball::generateRay + BvhInterface::intersect + getFinalColor on single rays
    cam = orc.make_camera(64, 48)
    for ndc in [(0.0, 0.0), (0.1, -0.2), (-0.9, 0.9)]:
        ray = srt.mirror_generate_ray(ndc, 64 / 48)
        assert_bits_equal(ray, orc.camera_rays(cam, [ndc])[0], "generateRay")
        want_hit, _ = ob.intersect(ray[None, :], feat)
        hit, t, info = srt.mirror_intersect_one(gb, ray, feat)
        assert hit == bool(want_hit["hit"][0])
        assert_bits_equal(np.float32(t), want_hit["t"][0], "ray.t")
        if hit:
            assert_bits_equal(info["normal"], want_hit["normal"][0], "HitInfo.normal")
            assert_bits_equal(info["uv"], want_hit["uv"][0], "HitInfo.texCoord")
            assert_bits_equal(info["kd"], want_hit["kd"][0], "HitInfo.material.kd (texel)")
            assert info["has_texture"] and abs(info["shininess"] - 250.0) < 1e-6
        want_rgb, _ = ob.final_color(ray[None, :], feat)
        assert np.abs(srt.mirror_final_color(gb, ray, feat) - want_rgb[0]).max() <= RGB_TOL
    # features outside the implemented path are refused loudly (no silent fallback)
    assert any(h for h, _, _ in [srt.mirror_intersect_one(gb, srt.mirror_generate_ray((0, 0), 64 / 48), feat)])


def test_frames_shared_between_scene_copies(gpu, scenes):
    """BvhInterface::addReplica + renderFrames: two more copies of the scene on the same GPU (a test box has one), three
    host threads drawing tiles of two cameras' frames from one counter.  Every frame is bit for bit the frame a single
    copy renders as one rectangle, whatever the tile size (ragged edge tiles included)."""
    d = dict(scenes("monkey"))
    feat = FEATURE_SETS["recursive"] | srt.F_MULTISAMPLE | srt.F_MISS_CULL | srt.F_COARSE_SHADOW
    cams = [(50.0, 3.0, 0, 0, 0, 20, 20, 0), (35.0, 4.0, 0.1, -0.1, 0.0, -30, 120, 10)]
    one = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)
    want = srt.mirror_render_frames(one, feat, 100, 70, cams, num_rays=2)
    many = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL)
    many.add_replica(0)
    many.add_replica(0)
    assert many.num_render_devices == 3 and one.num_render_devices == 1
    for tile in [(37, 30), (8, 8), (0, 0)]:
        got = srt.mirror_render_frames(many, feat, 100, 70, cams, num_rays=2, tile=tile)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), tile
    assert want[0].max() > 0.05 and not np.array_equal(want[0], want[1])
    # the single-frame entry point takes the same path
    a = srt.mirror_render(many, feat, 100, 70, 2)
    assert np.array_equal(a.view(np.uint32), want[0].view(np.uint32))


@pytest.mark.parametrize("name", ["cornell", "monkey", "dragon_standin"])
def test_device_tree_equals_the_oracle_tree(orc, scenes, gpu, name):
    """The node table as it sits on the device (srt_scene_download) against the oracle's (BoundingVolumeHierarchy's
    `nodes`, reference src/bounding_volume_hierarchy.cpp:14-94), with nothing of the product in between: walked side by
    side from the roots, every child box bit for bit, every leaf its triangles in the oracle's order, levels and counts."""
    if name == "dragon_standin":
        d = importlib.import_module("simple-ray-tracer_amd.standin").dragon_standin()
    else:
        d = scenes(name)
    ob = orc.Bvh(d, orc.F_ACCEL)
    got = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL).download()
    meta, box = ob.dump()  # per node: leaf, level, a, b (leaf: first entry / count of tri_order; inner: left / right child)
    order = ob.tri_order()
    assert (got["num_levels"], got["num_leaves"]) == (ob.num_levels, ob.num_leaves)
    root = len(meta) - 1
    assert got["root_box"].tobytes() == box[root].tobytes()
    nodes, tris = got["nodes"], got["tris"]
    LEAF = 0x80000000
    stack = [(got["root_ref"], root)]
    inner = leaves = 0
    while stack:
        ref, o = stack.pop()
        if ref & LEAF:
            first, cnt = ref & 0xFFFFFF, (ref >> 24) & 0x7F
            assert meta[o, 0] == 1 and cnt == meta[o, 3]
            want = order[meta[o, 2]:meta[o, 2] + cnt]
            assert np.array_equal(tris["mesh"][first:first + cnt], want[:, 0]) and np.array_equal(tris["tri"][first:first + cnt], want[:, 1])
            leaves += 1
            continue
        assert meta[o, 0] == 0
        n = nodes[ref]
        a, b = int(meta[o, 2]), int(meta[o, 3])
        assert n["lbox"].tobytes() == box[a].tobytes() and n["rbox"].tobytes() == box[b].tobytes(), (ref, o)
        assert int(n["left_level"]) == meta[a, 1] == meta[b, 1] == meta[o, 1] + 1
        inner += 1
        stack.append((int(n["left"]), a))
        stack.append((int(n["right"]), b))
    assert inner == len(nodes) and leaves == ob.num_leaves and inner + leaves == len(meta)


def test_uploaded_triangle_records_that_do_not_recompute_keep_the_full_records(orc, scenes, gpu, monkeypatch):
    """srt_scene_upload takes the 96-byte triangle records as given.  The coarse shadow walk's 48-byte records are only
    derived when the plane offset and the inside-test operands recomputed from normal and vertices equal the stored
    ones.  Here every other record gets its first operand negated (under the stored records' semantics those triangles
    can no longer be hit by any ray): the scene must keep the full records for its shadow walk, i.e. render the frame
    it renders with SRT_FULL_SHADOW_TRIANGLES set, and not the one of the untouched table."""
    import ctypes as C
    hip = C.CDLL(os.path.join(os.path.dirname(srt.__file__), "csrc", "libsrt_hip.so"))
    hip.srt_last_error.restype = C.c_char_p

    class Desc(C.Structure):
        _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_uint32), ("root_box", C.c_float * 6), ("root_ref", C.c_uint32),
                    ("num_levels", C.c_uint32), ("tris", C.c_void_p), ("attrs", C.c_void_p), ("n_tris", C.c_uint32),
                    ("scene_order", C.c_void_p), ("materials", C.c_void_p), ("n_materials", C.c_uint32), ("textures", C.c_void_p),
                    ("n_textures", C.c_uint32), ("texels", C.c_void_p), ("n_texels", C.c_uint64), ("lights", C.c_void_p),
                    ("n_lights", C.c_uint32), ("environment", C.c_uint32)]

    d = dict(scenes("monkey"))
    d["lights"] = np.concatenate([np.asarray(d["lights"], np.float32).reshape(-1, orc.LIGHT_STRIDE),
                                  np.asarray(orc.parallelogram_light((-0.3, 1.4, -0.3), (0.6, 0, 0), (0, 0, 0.6), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1)),
                                             np.float32).reshape(1, -1)])
    t = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL).download()
    mats = np.zeros((len(d["materials"]), 12), np.float32)
    mats[:, :8] = d["materials"]
    mats.view(np.int32)[:, 8] = -1
    lights = np.zeros((len(d["lights"]), 24), np.float32)
    lights[:, 1:22] = d["lights"][:, 1:22]
    lights.view(np.uint32)[:, 0] = d["lights"][:, 0].astype(np.uint32)
    cam = srt.make_camera(96, 72)
    feat = SH | srt.F_ACCEL | srt.F_HARD_SHADOW | srt.F_SOFT_SHADOW | srt.F_MISS_CULL | srt.F_COARSE_SHADOW

    def frame(tris):
        dev, scene = C.c_void_p(), C.c_void_p()
        assert hip.srt_device_create(0, C.byref(dev)) == 0, hip.srt_last_error()
        desc = Desc()
        desc.nodes, desc.n_nodes = t["nodes"].ctypes.data, len(t["nodes"])
        desc.root_box[:] = [float(x) for x in t["root_box"]]
        desc.root_ref, desc.num_levels = t["root_ref"], t["num_levels"]
        desc.tris, desc.attrs, desc.n_tris = tris.ctypes.data, t["attrs"].ctypes.data, len(tris)
        desc.scene_order = t["scene_order"].ctypes.data
        desc.materials, desc.n_materials = mats.ctypes.data, len(mats)
        desc.lights, desc.n_lights = lights.ctypes.data, len(lights)
        assert hip.srt_scene_upload(dev, C.byref(desc), C.byref(scene)) == 0, hip.srt_last_error()
        p = srt.RenderParams()
        p.camera, p.features, p.samples_x, p.samples_y, p.max_depth, p.seed = cam, feat, 1, 1, 5, 0
        p.rect[:] = [0, 0, cam.width, cam.height]
        rgb = np.zeros((cam.height, cam.width, 3), np.float32)
        out = srt.Output()
        out.rgb, out.out_width, out.out_height = rgb.ctypes.data, cam.width, cam.height
        assert hip.srt_render(dev, scene, C.byref(p), C.byref(out), None) == 0, hip.srt_last_error()
        hip.srt_scene_destroy(scene)
        hip.srt_device_destroy(dev)
        return rgb

    untouched = frame(t["tris"])
    holes = t["tris"].copy()
    holes["c0"][1::2] *= -1
    default = frame(holes)
    monkeypatch.setenv("SRT_FULL_SHADOW_TRIANGLES", "1")
    forced = frame(holes)
    monkeypatch.delenv("SRT_FULL_SHADOW_TRIANGLES")
    assert np.array_equal(default.view(np.uint32), forced.view(np.uint32))
    assert not np.array_equal(default.view(np.uint32), untouched.view(np.uint32))
    # and the untouched table is the frame of the device-built scene
    built = srt.Bvh(srt.Scene.from_arrays(d), srt.F_ACCEL).render(cam, feat)
    assert np.array_equal(untouched.view(np.uint32), built["rgb"].view(np.uint32))
