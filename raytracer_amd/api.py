"""ctypes binding of librt_hip.so (the C-ABI in include/rt_hip.h).

There is no CPU fallback: if the library is missing or a call fails, an
exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

from .scenes import LIGHT_ENV, Camera, Scene

# RT_HIP_LIB selects another build of the same library (kernel A/B runs)
_LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "librt_hip.so")
_lib = None


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rt_hip error {code}: {msg}")
        self.code = code


class rt_material(C.Structure):
    _fields_ = [("diffuse_color", C.c_float * 3), ("roughness", C.c_float), ("texture", C.c_int32),
                ("glossy", C.c_int32)]


class rt_texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgba8", C.POINTER(C.c_uint8))]


class rt_light(C.Structure):
    _fields_ = [("type", C.c_int32), ("pos", C.c_float * 3), ("intensity", C.c_float * 3),
                ("env_rgba", C.POINTER(C.c_float)), ("env_w", C.c_int32), ("env_h", C.c_int32),
                ("p0", C.c_float * 3), ("p1", C.c_float * 3), ("p2", C.c_float * 3)]


class rt_scene_desc(C.Structure):
    _fields_ = [("n_vertices", C.c_uint32),
                ("pos", C.POINTER(C.c_float)), ("pos_stride", C.c_uint32),
                ("normal", C.POINTER(C.c_float)), ("normal_stride", C.c_uint32),
                ("uv", C.POINTER(C.c_float)), ("uv_stride", C.c_uint32),
                ("tangent", C.POINTER(C.c_float)), ("tangent_stride", C.c_uint32),
                ("binormal", C.POINTER(C.c_float)), ("binormal_stride", C.c_uint32),
                ("n_indices", C.c_uint32), ("indices", C.POINTER(C.c_uint32)),
                ("tri_model", C.POINTER(C.c_int32)),
                ("n_models", C.c_uint32), ("materials", C.POINTER(rt_material)),
                ("n_lights", C.c_uint32), ("lights", C.POINTER(rt_light)),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(rt_texture))]


class rt_params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_depth", C.c_int32), ("tile_size", C.c_int32),
                ("seed", C.c_uint64), ("device", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32),
                ("flags", C.c_uint32), ("batch", C.c_uint32)]


RT_FLAG_HOST_BVH = 1
RT_FLAG_WIDE_BVH = 2
RT_NOALLOC_MAX_RAYS = 1 << 22   # rays per rt_trace_closest_noalloc call


class rt_stats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("passes", C.c_uint64),
                ("n_nodes", C.c_uint32), ("n_triangles", C.c_uint32), ("bvh_max_depth", C.c_int32),
                ("n_local_pixels", C.c_uint32), ("last_trace_ms", C.c_float), ("last_render_ms", C.c_float),
                ("bvh_build_ms", C.c_float), ("exact_closest_rays", C.c_uint64), ("exact_shadow_rays", C.c_uint64),
                ("device_bytes", C.c_uint64), ("path_state_bytes", C.c_uint64), ("batch", C.c_uint32)]


EXPORTS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_set_camera", "rt_reset_accumulation", "rt_render",
    "rt_read_film", "rt_film_slab", "rt_copy_film_slab", "rt_unpack_slabs", "rt_unpack_slabs_dev", "rt_trace_closest",
    "rt_trace_occluded", "rt_trace_closest_dev", "rt_trace_occluded_dev", "rt_get_bvh", "rt_get_stats",
    "rt_synchronize", "rt_version", "rt_lbvh_host", "rt_morton3d", "rt_last_trace_ms", "rt_slab_floats", "rt_pack_slab",
    "rt_read_primary_duv", "rt_trace_occluded_bounded", "rt_read_last_bounce_rays", "rt_trace_closest_noalloc",
    "rt_node_create", "rt_node_destroy", "rt_node_last_error", "rt_node_device_count", "rt_node_context", "rt_node_set_camera",
    "rt_node_reset_accumulation", "rt_node_render", "rt_node_gather_film", "rt_node_read_film", "rt_node_last_gather_ms",
    "rt_get_wide_bvh", "rt_node_map_film",
]


def load_library():
    """Loads librt_hip.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(f"{_LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
    # One HIP runtime per process: PyTorch ships its own libamdhip64, and a
    # process that loads the system one first and PyTorch's afterwards ends up
    # with no visible GPU in PyTorch.  Importing torch first makes this
    # library bind to the runtime torch uses (set RT_HIP_NO_TORCH=1 to skip).
    if os.environ.get("RT_HIP_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(_LIB_PATH)
    vp, fp, ip, up = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    lib.rt_create.argtypes = [C.POINTER(rt_scene_desc), C.POINTER(rt_params), C.POINTER(vp)]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_destroy.restype = None
    lib.rt_last_error.argtypes = [vp]
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_set_camera.argtypes = [vp, fp, fp, fp, fp]
    lib.rt_reset_accumulation.argtypes = [vp]
    lib.rt_render.argtypes = [vp, C.c_uint32]
    lib.rt_read_film.argtypes = [vp, fp]
    lib.rt_read_primary_duv.argtypes = [vp, fp]
    lib.rt_read_last_bounce_rays.argtypes = [vp, fp, C.c_uint32, up]
    lib.rt_film_slab.argtypes = [vp, C.POINTER(vp), up, C.POINTER(C.c_uint64)]
    lib.rt_copy_film_slab.argtypes = [vp, vp, C.c_uint64]
    lib.rt_unpack_slabs.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, C.c_uint64, fp]
    lib.rt_unpack_slabs_dev.argtypes = [vp, C.c_int32, vp, C.c_uint64, vp]
    lib.rt_trace_closest.argtypes = [vp, C.c_uint32, fp, fp, ip, ip, fp, fp]
    lib.rt_trace_occluded.argtypes = [vp, C.c_uint32, fp, ip]
    lib.rt_trace_occluded_bounded.argtypes = [vp, C.c_uint32, fp, fp, ip]
    lib.rt_trace_closest_dev.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, vp, C.c_uint32]
    lib.rt_trace_occluded_dev.argtypes = [vp, C.c_uint32, vp, vp, C.c_uint32]
    lib.rt_trace_closest_noalloc.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, vp]
    lib.rt_get_bvh.argtypes = [vp, up, fp, ip, ip, ip, ip, ip]
    lib.rt_get_wide_bvh.argtypes = [vp, up, ip, fp]
    lib.rt_get_stats.argtypes = [vp, C.POINTER(rt_stats)]
    lib.rt_synchronize.argtypes = [vp]
    lib.rt_version.restype = C.c_char_p
    lib.rt_lbvh_host.argtypes = [fp, C.c_uint32, up, C.c_uint32, ip, ip, fp, ip, ip, ip, ip, ip]
    lib.rt_morton3d.argtypes = [C.c_float, C.c_float, C.c_float]
    lib.rt_node_create.argtypes = [C.POINTER(rt_scene_desc), C.POINTER(rt_params), ip, C.c_int32, C.POINTER(vp)]
    lib.rt_node_destroy.argtypes = [vp]
    lib.rt_node_destroy.restype = None
    lib.rt_node_last_error.argtypes = [vp]
    lib.rt_node_last_error.restype = C.c_char_p
    lib.rt_node_device_count.argtypes = [vp]
    lib.rt_node_context.argtypes = [vp, C.c_int32]
    lib.rt_node_context.restype = vp
    lib.rt_node_set_camera.argtypes = [vp, fp, fp, fp, fp]
    lib.rt_node_reset_accumulation.argtypes = [vp]
    lib.rt_node_render.argtypes = [vp, C.c_uint32]
    lib.rt_node_gather_film.argtypes = [vp, C.POINTER(vp)]
    lib.rt_node_read_film.argtypes = [vp, fp]
    lib.rt_node_map_film.argtypes = [vp, C.POINTER(vp)]
    lib.rt_node_last_gather_ms.argtypes = [vp]
    lib.rt_node_last_gather_ms.restype = C.c_float
    for name in EXPORTS:
        if name not in ("rt_destroy", "rt_last_error", "rt_version", "rt_morton3d", "rt_last_trace_ms", "rt_slab_floats",
                        "rt_node_destroy", "rt_node_last_error", "rt_node_context", "rt_node_last_gather_ms"):
            getattr(lib, name).restype = C.c_int
    lib.rt_morton3d.restype = C.c_uint32
    lib.rt_slab_floats.restype = C.c_uint64
    lib.rt_slab_floats.argtypes = [C.c_int32] * 5
    lib.rt_pack_slab.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, fp]
    lib.rt_last_trace_ms.argtypes = [vp]
    lib.rt_last_trace_ms.restype = C.c_float
    _lib = lib
    return lib


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _iptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _scene_desc(s: Scene, use_frame: bool = True):
    """rt_scene_desc over the arrays of a Scene, plus the objects that must outlive the call."""
    keep = [s.pos, s.normal, s.uv, s.tangent, s.binormal, s.indices, s.tri_model]
    mats = (rt_material * s.n_models)()
    for m in range(s.n_models):
        mats[m].diffuse_color[:] = [float(x) for x in s.materials[m, :3]]
        mats[m].roughness = float(s.materials[m, 3])
        mats[m].texture = int(s.material_texture[m])
        mats[m].glossy = int(s.material_glossy[m])
    textures = (rt_texture * max(1, len(s.textures)))()
    for i, t in enumerate(s.textures):
        t = np.ascontiguousarray(t, np.uint8)
        keep.append(t)
        textures[i].height, textures[i].width = t.shape[0], t.shape[1]
        textures[i].rgba8 = t.ctypes.data_as(C.POINTER(C.c_uint8))
    lights = (rt_light * max(1, len(s.lights)))()
    for i, l in enumerate(s.lights):
        lights[i].type = l.type
        lights[i].pos[:] = [float(x) for x in l.pos]
        lights[i].intensity[:] = [float(x) for x in l.intensity]
        if l.tri is not None:
            for name, pt in zip(("p0", "p1", "p2"), l.tri):
                getattr(lights[i], name)[:] = [float(x) for x in pt]
        if l.type == LIGHT_ENV:
            env = np.ascontiguousarray(l.env, np.float32)
            keep.append(env)
            lights[i].env_rgba = _fptr(env)
            lights[i].env_h, lights[i].env_w = env.shape[0], env.shape[1]
    d = rt_scene_desc()
    d.n_vertices = s.pos.shape[0]
    d.pos, d.pos_stride = _fptr(s.pos), 12
    d.normal, d.normal_stride = _fptr(s.normal), 12
    d.uv, d.uv_stride = _fptr(s.uv), 8
    if use_frame:
        d.tangent, d.tangent_stride = _fptr(s.tangent), 12
        d.binormal, d.binormal_stride = _fptr(s.binormal), 12
    d.n_indices = s.indices.size
    d.indices = s.indices.ctypes.data_as(C.POINTER(C.c_uint32))
    d.tri_model = _iptr(s.tri_model)
    d.n_models, d.materials = s.n_models, mats
    d.n_lights, d.lights = len(s.lights), lights
    d.n_textures, d.textures = len(s.textures), textures
    keep += [mats, textures, lights]
    return d, keep


class Context:
    """One rt_ctx: a scene resident on one MI355X plus this rank's share of the film."""

    def __init__(self, scene: Scene, width: int, height: int, max_depth: int = 4, seed: int = 0, device: int = 0,
                 rank: int = 0, world: int = 1, tile_size: int = 64, use_frame: bool = True, host_bvh: bool = False, batch: int = 0,
                 wide_bvh=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.width, self.height, self.world, self.rank, self.tile_size = width, height, world, rank, tile_size
        d, self._keep = _scene_desc(scene, use_frame)
        p = rt_params(width, height, max_depth, tile_size, seed, device, rank, world, (RT_FLAG_HOST_BVH if host_bvh else 0) | (RT_FLAG_WIDE_BVH if _want_wide(wide_bvh) else 0), batch)
        rc = self._lib.rt_create(C.byref(d), C.byref(p), C.byref(self._h))
        if rc != 0:
            raise RtError(rc, self._lib.rt_last_error(None).decode())

    def _check(self, rc: int):
        if rc != 0:
            raise RtError(rc, self._lib.rt_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._lib.rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- integrator path
    def set_camera(self, cam: Camera):
        a = [np.ascontiguousarray(v, np.float32) for v in (cam.pos, cam.right, cam.up, cam.forward)]
        self._check(self._lib.rt_set_camera(self._h, *[_fptr(v) for v in a]))

    def reset_accumulation(self):
        self._check(self._lib.rt_reset_accumulation(self._h))

    def render(self, n_passes: int = 1):
        self._check(self._lib.rt_render(self._h, n_passes))

    def read_film(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._lib.rt_read_film(self._h, _fptr(out)))
        return out

    def read_primary_duv(self) -> np.ndarray:
        """(H, W, 4): dUVx.xy, dUVy.xy of the last pass's primary hits (zero where the ray missed)."""
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._lib.rt_read_primary_duv(self._h, _fptr(out)))
        return out

    def last_bounce_rays(self) -> np.ndarray:
        """(6, k) f32: the rays the last render() traced at its deepest bounce."""
        n = C.c_uint32()
        self._check(self._lib.rt_read_last_bounce_rays(self._h, None, 0, C.byref(n)))
        out = np.empty((6, n.value), np.float32)
        if n.value:
            self._check(self._lib.rt_read_last_bounce_rays(self._h, _fptr(out), n.value, C.byref(n)))
        return out

    def film_slab_info(self) -> Tuple[int, int, int]:
        ptr, nt, nf = C.c_void_p(), C.c_uint32(), C.c_uint64()
        self._check(self._lib.rt_film_slab(self._h, C.byref(ptr), C.byref(nt), C.byref(nf)))
        return int(ptr.value or 0), int(nt.value), int(nf.value)

    def copy_film_slab(self, device_ptr: int, n_floats: int):
        self._check(self._lib.rt_copy_film_slab(self._h, C.c_void_p(device_ptr), n_floats))

    def unpack_slabs_dev(self, world: int, slabs_device_ptr: int, per_rank_floats: int, film_device_ptr: int):
        """Scatters the all-gathered slabs (device memory) into a height*width*4 device film."""
        self._check(self._lib.rt_unpack_slabs_dev(self._h, world, C.c_void_p(slabs_device_ptr), per_rank_floats,
                                                  C.c_void_p(film_device_ptr)))

    # ---- BVH hooks
    def trace_closest(self, rays: np.ndarray):
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[1]
        t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32)
        id0 = np.empty(n, np.int32); id1 = np.empty(n, np.int32)
        self._check(self._lib.rt_trace_closest(self._h, n, _fptr(rays), _fptr(t), _iptr(id0), _iptr(id1), _fptr(u), _fptr(v)))
        return t, id0, id1, u, v

    def trace_occluded(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, np.float32)
        n = rays.shape[1]
        st = np.empty(n, np.int32)
        self._check(self._lib.rt_trace_occluded(self._h, n, _fptr(rays), _iptr(st)))
        return st

    def trace_occluded_bounded(self, rays: np.ndarray, tmax: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = rays.shape[1]
        assert tmax.shape == (n,)
        st = np.empty(n, np.int32)
        self._check(self._lib.rt_trace_occluded_bounded(self._h, n, _fptr(rays), _fptr(tmax), _iptr(st)))
        return st

    def trace_closest_dev(self, n: int, d_rays: int, d_t: int, d_id0: int, d_id1: int, d_u: int, d_v: int, repeat: int = 1):
        self._check(self._lib.rt_trace_closest_dev(self._h, n, d_rays, d_t, d_id0, d_id1, d_u, d_v, repeat))

    def trace_closest_noalloc(self, n: int, d_rays: int, d_t: int, d_id0: int, d_id1: int, d_u: int, d_v: int):
        """rt_trace_closest_noalloc: enqueues on the context's stream and returns; call synchronize() before reading."""
        self._check(self._lib.rt_trace_closest_noalloc(self._h, n, d_rays, d_t, d_id0, d_id1, d_u, d_v))

    def trace_occluded_dev(self, n: int, d_rays: int, d_status: int, repeat: int = 1):
        self._check(self._lib.rt_trace_occluded_dev(self._h, n, d_rays, d_status, repeat))

    def get_bvh(self):
        n = C.c_uint32()
        self._check(self._lib.rt_get_bvh(self._h, C.byref(n), None, None, None, None, None, None))
        n = n.value
        boxes = np.empty((n, 6), np.float32)
        arrs = [np.empty(n, np.int32) for _ in range(5)]
        nn = C.c_uint32()
        self._check(self._lib.rt_get_bvh(self._h, C.byref(nn), _fptr(boxes), *[_iptr(a) for a in arrs]))
        return (boxes, *arrs)

    def get_wide_bvh(self):
        """(nodes, depth): nodes is (n, 32) float32, the 4-wide nodes as rt_get_wide_bvh documents them."""
        n, d = C.c_uint32(), C.c_int32()
        self._check(self._lib.rt_get_wide_bvh(self._h, C.byref(n), C.byref(d), None))
        nodes = np.empty((n.value, 32), np.float32)
        if n.value:
            self._check(self._lib.rt_get_wide_bvh(self._h, C.byref(n), C.byref(d), _fptr(nodes)))
        return nodes, d.value

    def stats(self) -> dict:
        st = rt_stats()
        self._check(self._lib.rt_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in rt_stats._fields_}

    def last_trace_ms(self) -> float:
        return float(self._lib.rt_last_trace_ms(self._h))

    def synchronize(self):
        self._check(self._lib.rt_synchronize(self._h))


def _want_wide(wide_bvh) -> bool:
    """wide_bvh=None leaves the choice to the environment: RT_WIDE_BVH=1 makes every context that does not say
    otherwise walk the 4-wide tree (RT_FLAG_WIDE_BVH); the default is the reference's binary tree."""
    if wide_bvh is None:
        return os.environ.get("RT_WIDE_BVH", "0").lower() in ("1", "true", "yes")
    return bool(wide_bvh)


class Node:
    """One rt_node: several GPUs of a node driven by this process, one film sharded over them by interleaved tiles
    and gathered with RCCL at read-out (rt_node_* of include/rt_hip.h)."""

    def __init__(self, scene: Scene, width: int, height: int, devices, max_depth: int = 4, seed: int = 0, tile_size: int = 64,
                 use_frame: bool = True, batch: int = 0, wide_bvh=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.width, self.height, self.tile_size, self.devices = width, height, tile_size, list(devices)
        d, self._keep = _scene_desc(scene, use_frame)
        p = rt_params(width, height, max_depth, tile_size, seed, 0, 0, 1, RT_FLAG_WIDE_BVH if _want_wide(wide_bvh) else 0, batch)
        devs = (C.c_int32 * len(self.devices))(*self.devices)
        rc = self._lib.rt_node_create(C.byref(d), C.byref(p), devs, len(self.devices), C.byref(self._h))
        if rc != 0:
            raise RtError(rc, self._lib.rt_node_last_error(None).decode())

    def _check(self, rc: int):
        if rc != 0:
            raise RtError(rc, self._lib.rt_node_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._lib.rt_node_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_camera(self, cam: Camera):
        a = [np.ascontiguousarray(v, np.float32) for v in (cam.pos, cam.right, cam.up, cam.forward)]
        self._check(self._lib.rt_node_set_camera(self._h, *[_fptr(v) for v in a]))

    def reset_accumulation(self):
        self._check(self._lib.rt_node_reset_accumulation(self._h))

    def render(self, n_passes: int = 1):
        self._check(self._lib.rt_node_render(self._h, n_passes))

    def read_film(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._lib.rt_node_read_film(self._h, _fptr(out)))
        return out

    def map_film(self) -> np.ndarray:
        """The gathered film as a (H, W, 4) view of the node's pinned host buffer (valid until the next read-out)."""
        ptr = C.c_void_p()
        self._check(self._lib.rt_node_map_film(self._h, C.byref(ptr)))
        buf = (C.c_float * (self.height * self.width * 4)).from_address(ptr.value)
        return np.frombuffer(buf, np.float32).reshape(self.height, self.width, 4)

    def gather_film_device(self) -> int:
        """Device pointer (first device) of the gathered width*height*4 film."""
        ptr = C.c_void_p()
        self._check(self._lib.rt_node_gather_film(self._h, C.byref(ptr)))
        return int(ptr.value or 0)

    def last_gather_ms(self) -> float:
        return float(self._lib.rt_node_last_gather_ms(self._h))

    def device_stats(self, i: int) -> dict:
        st = rt_stats()
        h = self._lib.rt_node_context(self._h, i)
        rc = self._lib.rt_get_stats(h, C.byref(st))
        if rc != 0:
            raise RtError(rc, self._lib.rt_last_error(h).decode())
        return {k: getattr(st, k) for k, _ in rt_stats._fields_}


def unpack_slabs(width: int, height: int, tile_size: int, world: int, slabs: np.ndarray, per_rank: int) -> np.ndarray:
    lib = load_library()
    slabs = np.ascontiguousarray(slabs, np.float32)
    out = np.empty((height, width, 4), np.float32)
    rc = lib.rt_unpack_slabs(width, height, tile_size, world, _fptr(slabs), per_rank, _fptr(out))
    if rc != 0:
        raise RtError(rc, "rt_unpack_slabs failed")
    return out


def lbvh_host(scene: Scene):
    """Runs the library's host LBVH builder without a device: (max_depth,
    boxes, left, right, tri, model, leaf) in the reference's node numbering."""
    lib = load_library()
    n = 2 * scene.n_triangles - 1
    boxes = np.empty((n, 6), np.float32)
    arrs = [np.empty(n, np.int32) for _ in range(5)]
    md = C.c_int32()
    rc = lib.rt_lbvh_host(_fptr(scene.pos), 12, scene.indices.ctypes.data_as(C.POINTER(C.c_uint32)), scene.indices.size,
                          _iptr(scene.tri_model), C.byref(md), _fptr(boxes), *[_iptr(a) for a in arrs])
    if rc != 0:
        raise RtError(rc, "rt_lbvh_host failed")
    return (md.value, boxes, *arrs)


def slab_floats(width: int, height: int, tile_size: int, rank: int, world: int) -> int:
    return int(load_library().rt_slab_floats(width, height, tile_size, rank, world))


def pack_slab(film: np.ndarray, tile_size: int, rank: int, world: int) -> np.ndarray:
    """This rank's tiles of a (H, W, 4) film in slab order (host only)."""
    lib = load_library()
    film = np.ascontiguousarray(film, np.float32)
    h, w = film.shape[:2]
    out = np.empty(slab_floats(w, h, tile_size, rank, world), np.float32)
    rc = lib.rt_pack_slab(w, h, tile_size, rank, world, _fptr(film), _fptr(out))
    if rc != 0:
        raise RtError(rc, "rt_pack_slab failed")
    return out
