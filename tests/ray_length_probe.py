"""Development probe (CPU only; under tests/ because it runs the oracle): distribution of the traversal length (inner visits + triangle tests) of the rays of
the C2 sets, from the oracle.  The longest rays set the length of a launch's drain.
usage: python tests/ray_length_probe.py [stride]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import scenes  # noqa: E402
from oracle.binding import OracleScene  # noqa: E402

stride = int(sys.argv[1]) if len(sys.argv) > 1 else 4
s = scenes.sponza_class()
o = OracleScene(s)
L = o._L
L.orc_intersect_visits.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
for name, rays in (("coherent", scenes.camera_rays_numpy(s.camera, 1920, 1080)),
                   ("incoherent", scenes.incoherent_rays(s, 1920 * 1080, seed=0x5EED))):
    rays = np.ascontiguousarray(rays[:, ::stride])
    n = rays.shape[1]
    v = np.empty(n, np.uint32)
    L.orc_intersect_visits(o._h, n, rays.ctypes.data_as(C.POINTER(C.c_float)), v.ctypes.data_as(C.POINTER(C.c_uint32)))
    q = {p: float(np.percentile(v, p)) for p in (50, 90, 99, 99.9, 99.99)}
    print(f"{name:10s} n={n} mean {v.mean():.1f} percentiles {q} max {int(v.max())}")
