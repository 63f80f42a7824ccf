"""The GPU LBVH builder against the CPU oracle's tree (node for node, in the
reference's numbering) and against the library's own host builder."""
import numpy as np
import pytest

from oracle.binding import OracleScene
from raytracer_amd import Context, scenes
from tests.test_host_logic import _random_soup

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


MAKERS = {
    "cornell": lambda: scenes.cornell_box(),
    "atrium": lambda: scenes.sponza_class(detail=0.3, env_size=(16, 8)),
    "soup": lambda: _random_soup(2000, 1),
    "soup_duplicates": lambda: _random_soup(700, 2, dup=150),
    "soup_flat_axis": lambda: _random_soup(128, 3, flat=True),
    "one_triangle": lambda: _random_soup(1, 4),
    "two_triangles": lambda: _random_soup(2, 5),
}


@pytest.mark.parametrize("name", ["cornell", "atrium", "soup", "soup_duplicates", "soup_flat_axis", "one_triangle", "two_triangles"])
def test_gpu_built_tree_equals_oracle_and_host_builder(name):
    s = MAKERS[name]()
    dev = Context(s, 16, 16)                  # GPU builder (default)
    host = Context(s, 16, 16, host_bvh=True)  # host builder
    orc = OracleScene(s)
    got_d, got_h, ref = dev.get_bvh(), host.get_bvh(), orc.bvh_nodes()
    names = ["boxes", "left", "right", "tri", "model", "leaf"]
    for nm, d, h, r in zip(names, got_d, got_h, ref):
        if nm in ("left", "right"):      # children of leaves are unspecified (-1 here, stale in the reference)
            inner = ref[5] == 0
            d, h, r = d[inner], h[inner], r[inner]
        if nm in ("tri", "model"):
            lf = ref[5] == 1
            d, h, r = d[lf], h[lf], r[lf]
        assert np.array_equal(d, r), f"GPU-built {nm} differs from the oracle"
        assert np.array_equal(h, r), f"host-built {nm} differs from the oracle"
    assert dev.stats()["bvh_max_depth"] == host.stats()["bvh_max_depth"] == orc.bvh_info()[1]
    if s.n_triangles >= 2:
        rays = scenes.incoherent_rays(s, 20000, seed=3)
        a, b = dev.trace_closest(rays), host.trace_closest(rays)
        for x, y in zip(a, b):
            assert np.array_equal(_bits(x), _bits(y))
        assert np.array_equal(dev.trace_occluded(rays), host.trace_occluded(rays))
    dev.close()
    host.close()


def test_gpu_build_full_size_sponza_class_matches_host_builder():
    s = scenes.sponza_class(env_size=(16, 8))
    dev = Context(s, 16, 16)
    host = Context(s, 16, 16, host_bvh=True)
    for d, h in zip(dev.get_bvh(), host.get_bvh()):
        assert np.array_equal(d, h)
    sd, sh = dev.stats(), host.stats()
    assert sd["n_nodes"] == sh["n_nodes"] == 2 * s.n_triangles - 1
    print(f"LBVH build of {s.n_triangles} triangles: GPU {sd['bvh_build_ms']:.1f} ms, host {sh['bvh_build_ms']:.1f} ms")
    dev.close()
    host.close()
