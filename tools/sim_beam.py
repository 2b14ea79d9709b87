"""Development tool: CPU simulation of answering all samples of an area light with one walk of the ray bundle
(tools/sim_beam.cpp).  Prints the share of hits and of shadow-query work such a walk could take over."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

so = "/tmp/libsimbeam.so"
subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-fopenmp", "-shared", "-o", so,
                       os.path.join(ROOT, "tools", "sim_beam.cpp")])
import orc
orc._lib = None
_orig = C.CDLL


def _cdll(path, *a, **k):
    return _orig(so if path.endswith("liborc.so") else path, *a, **k)


C.CDLL = _cdll
L = orc.lib()
C.CDLL = _orig
L.sim_beam.argtypes = [C.c_void_p, C.POINTER(orc.Camera), C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_double)]

standin = importlib.import_module("simple-ray-tracer_amd.standin")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if name == "dragon":
    d = standin.dragon_standin(check=True)
    d["lights"] = standin.dragon_lights()
    light = 1
    W, H = 3840, 2160
else:
    d = orc.load_scene_npz(os.path.join(ROOT, "tests", "golden", "scenes", name + ".npz"))
    light = [i for i, l in enumerate(d["lights"]) if l[0] == 2][0]
    W, H = 1024, 1024
bvh = orc.Bvh(d, orc.F_ACCEL)
cam = orc.make_camera(W, H)
feat = orc.F_ACCEL | orc.F_SHADING | orc.F_SOFT_SHADOW | orc.F_HARD_SHADOW | orc.F_MISS_CULL | orc.F_COARSE_SHADOW
out = (C.c_double * 110)()
L.sim_beam(bvh.h, C.byref(cam), feat, light, step, out)
names = ["hits", "clean", "allvis", "allocc", "mixed", "cost_total", "cost_clean", "cost_allvis", "cost_allocc", "cost_mixed",
         "beam_visits", "beam_tris", "beam_visits_clean", "beam_tris_clean", "wrong", "fail_near", "fail_far", "fail_overflow",
         "vis_in_unclean", "q_total", "scout_work0", "scout_tried0", "scout_found0", "scout_work1", "scout_tried1", "scout_found1", "scout_work2", "scout_tried2", "scout_found2", "scout_work3", "scout_tried3", "scout_found3", "order_work0", "order_occ0", "order_work1", "order_occ1", "order_work2", "order_occ2", "order_work3", "order_occ3", "order_work4", "order_occ4", "order_work5", "order_occ5", "top_tests2", "top_roots2", "top_visits2", "top_tris2", "top_occ2", "top_tests3", "top_roots3", "top_visits3", "top_tris3", "top_occ3", "top_tests4", "top_roots4", "top_visits4", "top_tris4", "top_occ4", "fallbacks", "sah_visits0", "sah_tris0", "sah_occ0", "sah_visits1", "sah_tris1", "sah_occ1", "sah_visits2", "sah_tris2", "sah_occ2", "sah_visits3", "sah_tris3", "sah_occ3", "sah_visits4", "sah_tris4", "sah_occ4", "sah_visits5", "sah_tris5", "sah_occ5", "sah_maxsp", "sah_nodes", "zero_q", "zero_cost", "zero_vis", "vis_cost", "occ_cost"]
r = dict(zip(names, out))
for k in names:
    print(f"{k:20s} {r[k]:.0f}")
h = r["hits"]
print(f"hits: all visible {r['allvis']/h:.3f}, all occluded {r['allocc']/h:.3f}, mixed {r['mixed']/h:.3f}; clean bundles {r['clean']/h:.3f}")
print(f"work share: all visible {r['cost_allvis']/r['cost_total']:.3f}, all occluded {r['cost_allocc']/r['cost_total']:.3f}, "
      f"mixed {r['cost_mixed']/r['cost_total']:.3f}; clean bundles {r['cost_clean']/r['cost_total']:.3f}")
print(f"work per query {r['cost_total']/r['q_total']:.1f}; bundle walk per hit: {r['beam_visits']/h:.1f} wide visits + {r['beam_tris']/h:.1f} triangles "
      f"(clean ones: {r['beam_visits_clean']/max(r['clean'],1):.1f} + {r['beam_tris_clean']/max(r['clean'],1):.1f})")
print(f"queries whose Phong term is exactly zero: {r['zero_q']/r['q_total']:.3f} of all, {r['zero_cost']/r['cost_total']:.3f} of the work; {r['zero_vis']/max(r['zero_q'],1):.3f} of them visible")
print(f"work of visible queries {r['vis_cost']/r['cost_total']:.3f}, of occluded ones {r['occ_cost']/r['cost_total']:.3f}")
for k in range(4):
    print(f"scout, start {k} wide levels above the occluder: work {r[f'scout_work{k}']/r['cost_total']:.3f} of plain; tried on {r[f'scout_tried{k}']/r['q_total']:.3f} of the queries, found {r[f'scout_found{k}']/max(r[f'scout_tried{k}'],1):.3f}")
for k, nm in enumerate(["reference pop order", "nearest first", "farthest first", "nearest to the guide t (interval distance)", "longest span first", "nearest to the guide t (midpoint)"]):
    print(f"order {k} ({nm}): work {r[f'order_work{k}']/r['order_work0']:.3f} of the reference order's, occluded {r[f'order_occ{k}']:.0f}")
for k in (2, 3, 4):
    print(f"top {k} wide levels walked per hit by the bundle: {r[f'top_tests{k}']/h:.1f} box tests and {r[f'top_roots{k}']/h:.2f} listed sub-trees per hit; "
          f"per query {r[f'top_visits{k}']/r['q_total']:.2f} wide visits + {r[f'top_tris{k}']/r['q_total']:.2f} triangle tests "
          f"(work {(r[f'top_visits{k}']+r[f'top_tris{k}'])/r['order_work0']:.3f} of plain), occluded {r[f'top_occ{k}']:.0f}")
print(f"queries handed to the exact walk (stack capacity SRT_WIDE_CAP={os.environ.get('SRT_WIDE_CAP', 'default')}): {r['fallbacks']:.0f} of {r['q_total']:.0f} = {r['fallbacks']/r['q_total']:.5f}")
for k, nm in enumerate(["1 tri/leaf, left first", "1 tri/leaf, nearer first", "1 tri/leaf, farther first", "2 tri/leaf, left first", "2 tri/leaf, nearer first", "2 tri/leaf, farther first"]):
    print(f"SAH tree ({nm}): {r[f'sah_visits{k}']/r['q_total']:.2f} binary inner visits + {r[f'sah_tris{k}']/r['q_total']:.2f} triangle tests per query, occluded {r[f'sah_occ{k}']:.0f}")
print(f"SAH tree: {r['sah_nodes']:.0f} nodes, deepest stack {r['sah_maxsp']:.0f}; reference tree under miss culling: 35.2 inner visits + 7.4 triangle tests per query (GPU counters)")
