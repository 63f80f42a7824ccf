"""Development check (needs a -DRT_PROBE build): runs the deep-tree GPU test of the 4-wide traversal and prints how many
fix-up sections (LDS stack spill / refill) its launches executed, to confirm that the test exercises that path."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, '.')
import importlib.util
spec = importlib.util.spec_from_file_location("tw", "tests/test_gpu_wide.py")
from raytracer_amd import api
L = api.load_library()
L.rt_debug_probe.argtypes = [C.POINTER(C.c_uint64), C.c_int]
buf = (C.c_uint64 * 16)()
L.rt_debug_probe(buf, 1)
tw = importlib.util.module_from_spec(spec); spec.loader.exec_module(tw)
tw.test_wide_deep_tree_spills_and_refills_the_lds_stack()
L.rt_debug_probe(buf, 1)
v = [int(x) for x in buf]
print("fix-up sections", v[6] + v[14], "inner wave-steps", v[2] + v[10], "blocked lane-steps", v[7] + v[15])
