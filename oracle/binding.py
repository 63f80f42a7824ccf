"""ctypes binding of oracle/liboracle.so.  TEST INFRASTRUCTURE ONLY: imported
by tests/, the smoke check and bench.py's cpu_baseline leg; never by the
raytracer_amd package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_DIR, "liboracle.so")
_lib = None


class orc_light_desc(C.Structure):
    _fields_ = [("type", C.c_int32), ("pos", C.c_float * 3), ("intensity", C.c_float * 3),
                ("env_rgba", C.POINTER(C.c_float)), ("env_w", C.c_int32), ("env_h", C.c_int32),
                ("p0", C.c_float * 3), ("p1", C.c_float * 3), ("p2", C.c_float * 3)]


class orc_texture_desc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgba", C.POINTER(C.c_uint8))]


def build():
    subprocess.check_call(["make", "-C", _DIR, "-s"])


def set_math(mode: str):
    """"libm" (default: what the reference calls) or "shared" (the fixed polynomials the HIP kernels restate, under
    which films are comparable bit for bit).  Process-wide."""
    lib().orc_set_math({"libm": 0, "shared": 1}[mode])


def get_math() -> str:
    return "shared" if lib().orc_get_math() else "libm"


def dm_eval(which: str, x, y=None) -> np.ndarray:
    """Evaluates one of the shared-mode functions (sin, cos, acos, atan2(x, y), log2, log, exp, pow(x, y)) over float32 arrays."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(x if y is None else y, np.float32)
    out = np.empty_like(x)
    fp = C.POINTER(C.c_float)
    lib().orc_dm_eval(["sin", "cos", "acos", "atan2", "log2", "log", "exp", "pow"].index(which), x.size, x.ctypes.data_as(fp), y.ctypes.data_as(fp),
                      out.ctypes.data_as(fp))
    return out


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_PATH):
        build()
    L = C.CDLL(_PATH)
    vp, fp, ip, up = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    u64p = C.POINTER(C.c_uint64)
    L.orc_create.restype = vp
    L.orc_create.argtypes = [C.c_uint32, fp, fp, fp, fp, fp, C.c_uint32, up, ip, C.c_uint32, fp, C.c_uint32,
                             C.POINTER(orc_light_desc), ip, C.c_uint32, C.POINTER(orc_texture_desc), ip]
    L.orc_occluded_bounded.argtypes = [vp, C.c_uint64, fp, fp, ip]
    L.orc_set_math.argtypes = [C.c_int]
    L.orc_dm_eval.argtypes = [C.c_int, C.c_uint64, fp, fp, fp]
    for name in ("orc_erf_inv", "orc_erf_poly"):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = [C.c_float]
    L.orc_mf_D.restype = L.orc_mf_lambda.restype = L.orc_mf_pdf.restype = C.c_float
    L.orc_mf_D.argtypes = [C.c_int, C.c_float, C.c_float, fp]
    L.orc_mf_lambda.argtypes = [C.c_int, C.c_float, C.c_float, fp]
    L.orc_mf_pdf.argtypes = [C.c_int, C.c_float, C.c_float, fp, fp]
    L.orc_mf_sample_wh.argtypes = [C.c_int, C.c_float, C.c_float, fp, C.c_float, C.c_float, fp]
    L.orc_fresnel_conductor.argtypes = [fp, fp, fp, C.c_float, fp]
    L.orc_bsdf_f_pdf.restype = C.c_float
    L.orc_bsdf_f_pdf.argtypes = [C.c_int, C.c_float, fp, fp, fp, fp]
    L.orc_bsdf_sample.restype = C.c_float
    L.orc_bsdf_sample.argtypes = [C.c_int, C.c_float, fp, fp, C.c_float, C.c_float, fp, fp]
    L.orc_texture_levels.argtypes = [vp, C.c_int]
    L.orc_texture_level.argtypes = [vp, C.c_int, C.c_int, ip, ip, C.POINTER(C.c_uint8)]
    L.orc_texture_bilinear.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_float, fp]
    L.orc_texture_trilinear.argtypes = [vp, C.c_int, C.c_float, C.c_float, fp, fp, fp]
    L.orc_destroy.argtypes = [vp]
    L.orc_bvh_info.argtypes = [vp, up, ip]
    L.orc_bvh_nodes.argtypes = [vp, fp, ip, ip, ip, ip, ip]
    L.orc_intersect.argtypes = [vp, C.c_uint64, fp, fp, ip, ip, fp, fp, u64p, C.c_int]
    L.orc_occluded.argtypes = [vp, C.c_uint64, fp, ip, u64p]
    L.orc_intersect_ties.argtypes = [vp, C.c_uint64, fp, C.POINTER(C.c_uint8), C.c_int]
    L.orc_intersect_ties.restype = None
    L.orc_check_candidates.argtypes = [vp, C.c_uint64, fp, ip, fp, fp, fp, C.POINTER(C.c_uint8)]
    L.orc_check_candidates.restype = None
    L.orc_brute_force.argtypes = [vp, C.c_uint64, fp, fp, ip]
    L.orc_render.argtypes = [vp, fp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, fp, u64p,
                             C.c_int, fp]
    L.orc_capture_bounce_rays.argtypes = [vp, fp, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.c_int, fp,
                                          C.POINTER(C.c_uint8)]
    L.orc_camera_rays.argtypes = [fp, C.c_int, C.c_int, fp]
    L.orc_morton3d.restype = C.c_uint32
    L.orc_morton3d.argtypes = [C.c_float] * 3
    L.orc_expand_bits.restype = C.c_uint32
    L.orc_expand_bits.argtypes = [C.c_uint32]
    L.orc_intersect_bbox.argtypes = [fp, fp, fp, fp, fp]
    L.orc_intersect_triangle.argtypes = [fp, fp, fp, fp, fp, fp]
    L.orc_cosine_sample_hemisphere.argtypes = [C.c_float, C.c_float, fp]
    L.orc_diffuse_f.argtypes = [fp, C.c_float, fp, fp, fp]
    L.orc_diffuse_sample.restype = C.c_float
    L.orc_diffuse_sample.argtypes = [fp, C.c_float, fp, C.c_float, C.c_float, fp, fp]
    L.orc_power_heuristic.restype = C.c_float
    L.orc_power_heuristic.argtypes = [C.c_float, C.c_float]
    L.orc_dist1d_sample.restype = C.c_float
    L.orc_dist1d_sample.argtypes = [fp, C.c_int, C.c_float, fp, ip]
    L.orc_dist1d_cdf.argtypes = [fp, C.c_int, fp, fp]
    L.orc_dist2d_sample.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.c_float, fp, fp]
    L.orc_dist2d_pdf.restype = C.c_float
    L.orc_dist2d_pdf.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.c_float]
    L.orc_rng_u01.restype = C.c_float
    L.orc_rng_u01.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_light_sample.argtypes = [vp, C.c_int, fp, C.c_float, C.c_float, fp, fp, fp]
    L.orc_light_sample_t.restype = C.c_float
    L.orc_light_sample_t.argtypes = [vp, C.c_int, fp, C.c_float, C.c_float, fp, fp, fp]
    L.orc_light_pdf_at.restype = C.c_float
    L.orc_light_pdf_at.argtypes = [vp, C.c_int, fp, fp]
    L.orc_light_pdf.restype = C.c_float
    L.orc_light_pdf.argtypes = [vp, C.c_int, fp]
    L.orc_light_le.argtypes = [vp, C.c_int, fp, fp]
    _lib = L
    return L


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def f3(*v):
    return np.asarray(v, np.float32)


class OracleScene:
    """The CPU restatement of the reference path over one scene."""

    def __init__(self, scene, use_frame: bool = True):
        L = lib()
        self._L = L
        s = scene
        self._keep = []
        lights = (orc_light_desc * max(1, len(s.lights)))()
        for k, l in enumerate(s.lights):
            lights[k].type = l.type
            lights[k].pos[:] = [float(x) for x in l.pos]
            lights[k].intensity[:] = [float(x) for x in l.intensity]
            if l.env is not None:
                env = np.ascontiguousarray(l.env, np.float32)
                self._keep.append(env)
                lights[k].env_rgba = _f(env)
                lights[k].env_h, lights[k].env_w = env.shape[0], env.shape[1]
            if getattr(l, "tri", None) is not None:
                for name, pt in zip(("p0", "p1", "p2"), l.tri):
                    getattr(lights[k], name)[:] = [float(x) for x in pt]
        mats = np.ascontiguousarray(s.materials, np.float32)
        textures = list(getattr(s, "textures", []))
        tex = (orc_texture_desc * max(1, len(textures)))()
        for k, t in enumerate(textures):
            t = np.ascontiguousarray(t, np.uint8)
            self._keep.append(t)
            tex[k].height, tex[k].width = t.shape[0], t.shape[1]
            tex[k].rgba = t.ctypes.data_as(C.POINTER(C.c_uint8))
        mt = getattr(s, "material_texture", None)
        mt = np.full(s.n_models, -1, np.int32) if mt is None else np.ascontiguousarray(mt, np.int32)
        mg = getattr(s, "material_glossy", None)
        mg = np.zeros(s.n_models, np.int32) if mg is None else np.ascontiguousarray(mg, np.int32)
        self._h = L.orc_create(s.pos.shape[0], _f(s.pos), _f(s.normal), _f(s.uv),
                               _f(s.tangent) if use_frame else None, _f(s.binormal) if use_frame else None,
                               s.indices.size, s.indices.ctypes.data_as(C.POINTER(C.c_uint32)), _i(s.tri_model),
                               s.n_models, _f(mats), len(s.lights), lights, _i(mt), len(textures), tex, _i(mg))
        self.scene = s

    def __del__(self):
        try:
            if self._h:
                self._L.orc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def bvh_info(self):
        n, d = C.c_uint32(), C.c_int32()
        self._L.orc_bvh_info(self._h, C.byref(n), C.byref(d))
        return n.value, d.value

    def bvh_nodes(self):
        n, _ = self.bvh_info()
        boxes = np.empty((n, 6), np.float32)
        arrs = [np.empty(n, np.int32) for _ in range(5)]
        self._L.orc_bvh_nodes(self._h, _f(boxes), *[_i(a) for a in arrs])
        return (boxes, *arrs)

    def intersect(self, rays, counters=False, threads=1):
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[1]
        t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32)
        id0 = np.empty(n, np.int32); id1 = np.empty(n, np.int32)
        cnt = (C.c_uint64 * 5)()
        self._L.orc_intersect(self._h, n, _f(rays), _f(t), _i(id0), _i(id1), _f(u), _f(v), cnt if counters else None, threads)
        if counters:
            return t, id0, id1, u, v, {"n_inner": cnt[0], "n_tri": cnt[1], "n_rays": cnt[2], "n_inner_stale": cnt[3], "n_ties": cnt[4]}
        return t, id0, id1, u, v

    def intersect_ties(self, rays, threads=1):
        """uint8 per ray: 1 when two or more reachable triangles lie at exactly the distance of the closest hit (the id the
        strict `<` of src/scene/BVH.cpp:139 keeps then depends on the visiting order)."""
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[1]
        flags = np.empty(n, np.uint8)
        self._L.orc_intersect_ties(self._h, n, _f(rays), flags.ctypes.data_as(C.POINTER(C.c_uint8)), threads)
        return flags

    def check_candidates(self, rays, id1):
        """For hits another visiting order reported: (t, u, v, ok) of the reference's triangle test on triangle id1[i];
        ok bit 0 = the test passes with t > 1e-4, bit 1 = every box from the root to the triangle's leaf passes."""
        rays = np.ascontiguousarray(rays, np.float32)
        id1 = np.ascontiguousarray(id1, np.int32)
        n = rays.shape[1]
        t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32)
        ok = np.empty(n, np.uint8)
        self._L.orc_check_candidates(self._h, n, _f(rays), _i(id1), _f(t), _f(u), _f(v), ok.ctypes.data_as(C.POINTER(C.c_uint8)))
        return t, u, v, ok

    def occluded(self, rays):
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[1]
        st = np.empty(n, np.int32)
        self._L.orc_occluded(self._h, n, _f(rays), _i(st), None)
        return st

    def occluded_bounded(self, rays, tmax):
        rays = np.ascontiguousarray(rays, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        st = np.empty(rays.shape[1], np.int32)
        self._L.orc_occluded_bounded(self._h, rays.shape[1], _f(rays), _f(tmax), _i(st))
        return st

    def brute_force(self, rays):
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[1]
        t = np.empty(n, np.float32); id1 = np.empty(n, np.int32)
        self._L.orc_brute_force(self._h, n, _f(rays), _f(t), _i(id1))
        return t, id1

    def render(self, cam, W, H, max_depth, seed, n_passes, threads=8, film=None, first_pass=0, tile_size=64, duv=None):
        """duv: optional (H, W, 4) float32 array receiving dUVx, dUVy of each primary hit."""
        if film is None:
            film = np.zeros((H, W, 4), np.float32)
        cnt = (C.c_uint64 * 12)()
        c = cam.packed()
        self._L.orc_render(self._h, _f(c), W, H, max_depth, seed, first_pass, n_passes, threads, _f(film), cnt, tile_size,
                            None if duv is None else _f(duv))
        keys = ("closest", "shadow", "closest_inner", "closest_tri", "shadow_inner", "shadow_tri", "prim_rays", "prim_inner", "prim_tri",
                "prim_point_rays", "prim_point_inner", "prim_point_tri")
        return film, {k: int(cnt[i]) for i, k in enumerate(keys)}

    def bounce_rays(self, cam, W, H, n, depth=0, max_depth=4, seed=0, threads=8):
        """(6, n) f32: bounce rays leaving the `depth`-th shaded vertex, gathered pass by pass until n are found."""
        out, have, p = np.empty((6, n), np.float32), 0, 0
        rays = np.empty((6, W * H), np.float32)
        flags = np.empty(W * H, np.uint8)
        c = cam.packed()
        while have < n:
            self._L.orc_capture_bounce_rays(self._h, _f(c), W, H, max_depth, seed, p, threads, depth, _f(rays),
                                            flags.ctypes.data_as(C.POINTER(C.c_uint8)))
            idx = np.nonzero(flags)[0][: n - have]
            if idx.size == 0 and p > 64:
                raise RuntimeError("the scene produces no bounce rays")
            out[:, have:have + idx.size] = rays[:, idx]
            have += idx.size
            p += 1
        return out

    def texture_level(self, t, level):
        w, h = C.c_int32(), C.c_int32()
        self._L.orc_texture_level(self._h, t, level, C.byref(w), C.byref(h), None)
        out = np.empty((h.value, w.value, 4), np.uint8)
        self._L.orc_texture_level(self._h, t, level, C.byref(w), C.byref(h), out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out

    def texture_levels(self, t):
        return self._L.orc_texture_levels(self._h, t)

    def texture_bilinear(self, t, level, u, v):
        rgb = np.zeros(3, np.float32)
        self._L.orc_texture_bilinear(self._h, t, level, u, v, _f(rgb))
        return rgb

    def texture_trilinear(self, t, u, v, duvx, duvy):
        rgb = np.zeros(3, np.float32)
        self._L.orc_texture_trilinear(self._h, t, u, v, _f(f3(*duvx)), _f(f3(*duvy)), _f(rgb))
        return rgb

    def light_sample(self, l, p, u0, u1):
        in3, li3 = np.zeros(3, np.float32), np.zeros(3, np.float32)
        pdf = C.c_float()
        self._L.orc_light_sample(self._h, l, _f(f3(*p)), u0, u1, _f(in3), C.byref(pdf), _f(li3))
        return in3, pdf.value, li3

    def light_sample_t(self, l, p, u0, u1):
        """(in, pdf, li, tmax): sampleLi with the far bound of its shadow ray."""
        in3, li3 = np.zeros(3, np.float32), np.zeros(3, np.float32)
        pdf = C.c_float()
        t = self._L.orc_light_sample_t(self._h, l, _f(f3(*p)), u0, u1, _f(in3), C.byref(pdf), _f(li3))
        return in3, pdf.value, li3, t

    def light_pdf_at(self, l, p, d):
        return self._L.orc_light_pdf_at(self._h, l, _f(f3(*p)), _f(f3(*d)))

    def light_pdf(self, l, d):
        return self._L.orc_light_pdf(self._h, l, _f(f3(*d)))

    def light_le(self, l, d):
        le = np.zeros(3, np.float32)
        self._L.orc_light_le(self._h, l, _f(f3(*d)), _f(le))
        return le


def camera_rays(cam, W, H):
    rays = np.empty((6, W * H), np.float32)
    c = cam.packed()
    lib().orc_camera_rays(_f(c), W, H, _f(rays))
    return rays
