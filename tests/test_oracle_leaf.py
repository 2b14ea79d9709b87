"""The CPU oracle against golden vectors produced by the reference's own object code
(oracle/make_golden.py): every leaf function must agree bit for bit."""
import ctypes as C

import numpy as np

from conftest import assert_bits_equal

f32p = C.POINTER(C.c_float)


def p(a):
    return a.ctypes.data_as(f32p)


def test_ray_triangle(orc, leaf_golden):
    g = leaf_golden
    L = orc.lib()
    tri, rays = g["tri_tri9"], g["tri_rays"].copy()
    hit = np.zeros(len(rays), np.int32)
    for i in range(len(rays)):
        hit[i] = L.orc_ray_triangle(p(tri[i, 0:3]), p(tri[i, 3:6]), p(tri[i, 6:9]), p(rays[i]))
    assert np.array_equal(hit, g["tri_hit"])
    assert_bits_equal(rays[:, 6], g["tri_t"], "ray/triangle t")
    assert 1000 < hit.sum() < len(hit) - 1000


def test_ray_box(orc, leaf_golden):
    g = leaf_golden
    L = orc.lib()
    box, rays = g["box_box6"], g["box_rays"].copy()
    hit = np.zeros(len(rays), np.int32)
    for i in range(len(rays)):
        hit[i] = L.orc_ray_aabb(p(box[i]), p(rays[i]))
    assert np.array_equal(hit, g["box_hit"])
    assert_bits_equal(rays[:, 6], g["box_t"], "ray/box t")


def test_barycentric(orc, leaf_golden):
    g = leaf_golden
    L = orc.lib()
    tri, P = g["bary_tri9"], g["bary_p"]
    out = np.zeros((len(P), 3), np.float32)
    for i in range(len(P)):
        L.orc_barycentric(p(tri[i, 0:3]), p(tri[i, 3:6]), p(tri[i, 6:9]), p(P[i]), p(out[i]))
    assert_bits_equal(out, g["bary_out"], "barycentric")
    assert (out[20:40] == [1, 0, 0]).all()


def test_interpolation_formulas(leaf_golden):
    # interpolateNormal / interpolateTexCoord are ((b.x*a0 + b.y*a1) + b.z*a2) in float32;
    # numpy float32 arithmetic reproduces them, which fixes the evaluation order the oracle uses
    g = leaf_golden
    b = g["bary_out"]
    n9 = g["interp_n9"].reshape(-1, 3, 3)
    n = (b[:, 0:1] * n9[:, 0] + b[:, 1:2] * n9[:, 1]) + b[:, 2:3] * n9[:, 2]
    assert_bits_equal(n, g["interp_n_out"], "interpolateNormal")
    t6 = g["interp_t6"].reshape(-1, 3, 2)
    t = (t6[:, 0] * b[:, 0:1] + t6[:, 1] * b[:, 1:2]) + t6[:, 2] * b[:, 2:3]
    assert_bits_equal(t, g["interp_t_out"], "interpolateTexCoord")


def test_texel(orc, leaf_golden, scenes):
    g = leaf_golden
    L = orc.lib()
    d = scenes("cube_textured")
    w, h = int(d["tex_w"][0]), int(d["tex_h"][0])
    out = np.zeros((len(g["texel_uv"]), 3), np.float32)
    for i, (u, v) in enumerate(g["texel_uv"]):
        L.orc_texel(p(d["texels"]), w, h, float(u), float(v), p(out[i]))
    assert_bits_equal(out, g["texel_out"], "acquireTexel")


def test_shading(orc, leaf_golden):
    g = leaf_golden
    L = orc.lib()
    n = len(g["shade_rays"])
    out = np.zeros((n, 3), np.float32)
    for i in range(n):
        L.orc_compute_shading(p(g["shade_lp"][i]), p(g["shade_lc"][i]), p(g["shade_rays"][i]), p(g["shade_n"][i]),
                              p(g["shade_kd"][i]), p(g["shade_ks"][i]), float(g["shade_sh"][i]), p(out[i]))
    # same libm (powf) on both sides in this container -> bit equal; elsewhere allow libm's last ulps
    ref = g["shade_out"]
    assert np.allclose(out, ref, rtol=2e-6, atol=1e-7)


def test_reflection_ray(orc, leaf_golden):
    g = leaf_golden
    L = orc.lib()
    n = len(g["shade_rays"])
    out = np.zeros((n, 7), np.float32)
    for i in range(n):
        L.orc_reflection_ray(p(g["shade_rays"][i]), p(g["shade_n"][i]), p(out[i]))
    assert_bits_equal(out, g["refl_out"], "computeReflectionRay")


def test_reflection_offset_constant():
    # glm::pow(10, -5) narrowed to float (reference src/shading.cpp:40)
    assert np.float32(10.0 ** -5).view(np.uint32) == 0x3727C5AC


def test_camera(orc, camera_golden):
    g = camera_golden
    for i, cfg in enumerate(g["cfg"]):
        fov, dist, la, rot, w, h = cfg[0], cfg[1], cfg[2:5], cfg[5:8], int(cfg[8]), int(cfg[9])
        cam = orc.make_camera(w, h, fov, dist, la, rot)
        # sin/cos/tan come from libm: identical here, last-ulp freedom elsewhere
        assert np.allclose(np.array(cam.quat[:]), g[f"quat{i}"], rtol=1e-6, atol=1e-7)
        assert np.allclose([cam.half_w, cam.half_h], g[f"half{i}"], rtol=1e-6)
        assert np.allclose(np.array(cam.origin[:]), g[f"origin{i}"], rtol=1e-6, atol=1e-6)
        # with the golden camera constants plugged in, ray generation has no libm call left: bit exact
        cam.quat[:] = g[f"quat{i}"].tolist()
        cam.origin[:] = g[f"origin{i}"].tolist()
        cam.half_w, cam.half_h = float(g[f"half{i}"][0]), float(g[f"half{i}"][1])
        rays = orc.camera_rays(cam, g[f"ndc{i}"])
        assert_bits_equal(rays, g[f"rays{i}"], f"camera rays cfg {i}")
