"""CPU-side checks of the product's host code: the C-ABI library loads and
exports what include/rt_hip.h declares, argument validation, the host LBVH
builder against the oracle's tree, and the tile-slab pack/unpack pair."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import binding as ob
from raytracer_amd import api, scenes
from tests.test_oracle_kats import soup

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = api.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in rt_hip.h but not exported"
    assert set(api.EXPORTS) == declared
    assert b"gfx950" in lib.rt_version()


def test_create_rejects_bad_arguments_before_touching_a_device():
    lib = api.load_library()
    h = C.c_void_p()
    assert lib.rt_create(None, None, C.byref(h)) == -1
    d, p = api.rt_scene_desc(), api.rt_params()
    assert lib.rt_create(C.byref(d), C.byref(p), C.byref(h)) == -1
    assert b"positions" in lib.rt_last_error(None)
    assert not h.value


def test_create_without_gpu_fails_loudly(cornell):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.RtError) as e:
        api.Context(cornell, 32, 32)
    assert e.value.code == -3


def test_node_create_argument_errors(cornell):
    lib = api.load_library()
    h = C.c_void_p()
    d, p = api.rt_scene_desc(), api.rt_params()
    devs = (C.c_int32 * 2)(0, 0)
    assert lib.rt_node_create(None, None, devs, 1, C.byref(h)) == -1
    assert lib.rt_node_create(C.byref(d), C.byref(p), devs, 0, C.byref(h)) == -1
    assert lib.rt_node_create(C.byref(d), C.byref(p), devs, 2, C.byref(h)) == -1 and b"twice" in lib.rt_node_last_error(None)
    assert not h.value
    assert lib.rt_node_render(None, 1) == -1 and lib.rt_node_read_film(None, None) == -1
    assert lib.rt_node_device_count(None) == 0 and not lib.rt_node_context(None, 0)
    lib.rt_node_destroy(None)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(api.RtError) as e:
            api.Node(cornell, 32, 32, [0])
        assert e.value.code == -3 and "device 0" in str(e.value)


def test_null_context_calls_return_arg_error():
    lib = api.load_library()
    assert lib.rt_render(None, 1) == -1
    assert lib.rt_get_stats(None, None) == -1
    assert lib.rt_read_film(None, None) == -1
    lib.rt_destroy(None)  # no-op


def _random_soup(n, seed, dup=0, flat=False):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-5, 5, (n, 1, 3))
    if flat:
        c[..., 1] = 0.0
    tris = c + rng.uniform(-0.3, 0.3, (n, 3, 3)) * (np.array([1, 0, 1]) if flat else 1)
    tris = [tuple(map(tuple, t)) for t in tris]
    tris += tris[:dup]  # exact duplicates: identical Morton codes
    return soup(tris, models=list(rng.integers(0, 3, len(tris))))


@pytest.mark.parametrize("maker", [
    lambda: scenes.cornell_box(),
    lambda: scenes.sponza_class(detail=0.3, env_size=(16, 8)),
    lambda: _random_soup(500, 1),
    lambda: _random_soup(300, 2, dup=40),
    lambda: _random_soup(64, 3, flat=True),   # flat axis: 0/0 in the Morton normalisation
    lambda: _random_soup(1, 4),
    lambda: _random_soup(2, 5),
])
def test_host_lbvh_matches_oracle_tree(maker):
    s = maker()
    md, boxes, left, right, tri, model, leaf = api.lbvh_host(s)
    o = ob.OracleScene(s)
    n, depth = o.bvh_info()
    rb, rl, rr, rtri, rmodel, rleaf = o.bvh_nodes()
    assert n == 2 * s.n_triangles - 1 == len(left)
    assert md == depth
    assert np.array_equal(leaf, rleaf)
    inner = leaf == 0
    assert np.array_equal(left[inner], rl[inner]) and np.array_equal(right[inner], rr[inner])
    assert np.array_equal(tri[~inner], rtri[~inner]) and np.array_equal(model[~inner], rmodel[~inner])
    assert np.array_equal(boxes, rb)  # == on floats: -0 and +0 compare equal
    # every triangle is in exactly one leaf
    assert sorted(tri[~inner].tolist()) == list(range(0, 3 * s.n_triangles, 3))


def test_morton_matches_oracle():
    lib, L = api.load_library(), ob.lib()
    rng = np.random.default_rng(0)
    for x, y, z in rng.uniform(-0.2, 1.2, (2000, 3)).astype(np.float32):
        assert lib.rt_morton3d(x, y, z) == L.orc_morton3d(x, y, z)
    assert lib.rt_morton3d(float("nan"), 0.5, 0.5) == L.orc_morton3d(float("nan"), 0.5, 0.5)


@pytest.mark.parametrize("w,h,tile,world", [(100, 70, 64, 1), (100, 70, 64, 2), (130, 130, 32, 3), (64, 64, 64, 4),
                                             (1920, 1080, 64, 8), (17, 9, 8, 2)])
def test_slab_pack_unpack_round_trip(w, h, tile, world):
    rng = np.random.default_rng(w * h + world)
    film = rng.random((h, w, 4), np.float32)
    sizes = [api.slab_floats(w, h, tile, r, world) for r in range(world)]
    n_tiles = -(-w // tile) * -(-h // tile)
    assert sum(sizes) == n_tiles * tile * tile * 4
    per = max(sizes)
    slabs = np.zeros((world, per), np.float32)
    for r in range(world):
        slabs[r, : sizes[r]] = api.pack_slab(film, tile, r, world)
    out = api.unpack_slabs(w, h, tile, world, slabs, per)
    assert np.array_equal(out, film)
    # a rank's slab holds only its own tiles: zeroing another rank's slab
    # leaves this rank's pixels intact
    if world > 1:
        slabs2 = slabs.copy()
        slabs2[1] = 0
        out2 = api.unpack_slabs(w, h, tile, world, slabs2, per)
        ty, tx = np.mgrid[0:h, 0:w]
        owner = ((ty // tile) * -(-w // tile) + tx // tile) % world
        assert np.array_equal(out2[owner != 1], film[owner != 1])
        assert not out2[owner == 1].any()


def test_incoherent_rays_are_seeded_and_unit_length(cornell):
    a = scenes.incoherent_rays(cornell, 1000, seed=3)
    b = scenes.incoherent_rays(cornell, 1000, seed=3)
    c = scenes.incoherent_rays(cornell, 1000, seed=4)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.allclose(np.linalg.norm(a[3:], axis=0), 1.0, atol=1e-5)
    lo, hi = cornell.pos.min(0), cornell.pos.max(0)
    assert (a[:3].T >= lo - 1e-6).all() and (a[:3].T <= hi + 1e-6).all()


def test_camera_rays_numpy_equals_oracle(cornell):
    for w, h in ((64, 64), (33, 17)):
        assert np.array_equal(scenes.camera_rays_numpy(cornell.camera, w, h), ob.camera_rays(cornell.camera, w, h))
    cam = scenes.make_camera((1, 2, 3), (0.3, -0.2, 0.9))
    assert np.array_equal(scenes.camera_rays_numpy(cam, 40, 30), ob.camera_rays(cam, 40, 30))


def test_sponza_class_shape():
    s = scenes.sponza_class(env_size=(16, 8))
    assert abs(s.n_triangles - 262144) / 262144 < 0.05
    ext = s.pos.max(0) - s.pos.min(0)
    assert ext == pytest.approx([37, 15, 23], abs=0.6)
    assert 15 <= s.n_models <= 30
    assert s.tri_model.max() == s.n_models - 1


class _StubContext:
    """Stands in for api.Context in the adapter protocol tests: render() fails from the n-th call on."""

    def __init__(self, fail_from=1):
        self.calls, self.fail_from = 0, fail_from

    def set_camera(self, cam):
        pass

    def render(self, n):
        self.calls += 1
        if self.calls >= self.fail_from:
            raise api.RtError(-4, "forced")

    def read_film(self):
        return np.zeros((4, 4, 4), np.float32)

    def close(self):
        pass


def _stub_adapter(fail_from):
    from raytracer_amd.integrator import HipPathTracer
    pt = object.__new__(HipPathTracer)
    pt._passes, pt._ctx = 1, _StubContext(fail_from)
    pt._width = pt._height = 4
    pt._frame = np.zeros((4, 4, 4), np.float32)
    pt._texture = np.zeros((4, 4, 4), np.float32)
    pt._isFinish, pt._isRun, pt._camera, pt._thread, pt._frames, pt.error = True, True, None, None, 0, None
    return pt


def _wait(pred, seconds=5.0):
    import time
    t0 = time.time()
    while not pred() and time.time() - t0 < seconds:
        time.sleep(0.001)
    return pred()


def test_adapter_close_returns_after_a_render_error(cornell):
    pt = _stub_adapter(fail_from=2)
    pt.nextFrame(cornell.camera)
    pt.run()
    assert _wait(pt.isUpdateFinish)
    assert pt.error is None and pt.frames == 1
    pt.nextFrame(cornell.camera)          # this update fails
    assert _wait(lambda: pt.error is not None)
    assert _wait(pt.isUpdateFinish)
    pt.nextFrame(cornell.camera)          # ignored after an error: the frame stays finished
    assert pt.isUpdateFinish()
    pt.close()                            # must not hang
    assert not pt._thread.is_alive()


def test_adapter_finish_right_after_next_frame_returns(cornell):
    pt = _stub_adapter(fail_from=10 ** 9)
    pt.run()
    pt.nextFrame(cornell.camera)
    pt.finish()                           # whether or not the thread picked the frame up
    assert pt.isUpdateFinish() and not pt._thread.is_alive()
    pt.nextFrame(cornell.camera)          # no thread left: ignored
    assert pt.isUpdateFinish()


def test_cpp_adapter_shutdown_with_a_failing_library(tmp_path):
    """HipPathTracer.hpp against a stub of the C-ABI whose rt_render fails on the second call: error() is set,
    nextFrame() is ignored afterwards, finish() and the destructor return."""
    import subprocess
    stub = tmp_path / "stub.cpp"
    stub.write_text(r'''
#include "raytracer_amd/host/HipPathTracer.hpp"
#include <chrono>
#include <cstdio>
static int g_calls = 0;
extern "C" {
int rt_create(const rt_scene_desc*, const rt_params*, rt_ctx** out) { *out = reinterpret_cast<rt_ctx*>(&g_calls); return 0; }
void rt_destroy(rt_ctx*) {}
const char* rt_last_error(const rt_ctx*) { return "forced failure"; }
int rt_set_camera(rt_ctx*, const float*, const float*, const float*, const float*) { return 0; }
int rt_render(rt_ctx*, uint32_t) { return ++g_calls >= 2 ? RT_ERR_LIMIT : RT_OK; }
int rt_read_film(rt_ctx*, float*) { return 0; }
int rt_node_create(const rt_scene_desc*, const rt_params*, const int32_t*, int32_t, rt_node**) { return RT_ERR_ARG; }
void rt_node_destroy(rt_node*) {}
const char* rt_node_last_error(const rt_node*) { return ""; }
int rt_node_device_count(const rt_node*) { return 0; }
rt_ctx* rt_node_context(rt_node*, int32_t) { return nullptr; }
int rt_node_set_camera(rt_node*, const float*, const float*, const float*, const float*) { return 0; }
int rt_node_render(rt_node*, uint32_t) { return 0; }
int rt_node_read_film(rt_node*, float*) { return 0; }
}
static bool wait_finish(HipPathTracer& pt) {
    auto t0 = std::chrono::steady_clock::now();
    while (!pt.isUpdateFinish())
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) return false;
    return true;
}
int main() {
    rt_scene_desc sd{};
    const float pos[3] = {0, 0, 0}, right[3] = {1, 0, 0}, fwd[3] = {0, 0, -1};
    {
        HipPathTracer pt(sd, 4, 4);
        pt.nextFrame(pos, right, fwd);
        pt.run();
        if (!wait_finish(pt) || pt.failed()) return 1;
        pt.nextFrame(pos, right, fwd);
        if (!wait_finish(pt) || !pt.failed() || pt.error() != "forced failure") return 2;
        pt.nextFrame(pos, right, fwd);
        if (!pt.isUpdateFinish()) return 3;
        pt.finish();
    }
    {
        HipPathTracer pt(sd, 4, 4);
        pt.run();
        pt.nextFrame(pos, right, fwd);
        pt.finish();
        if (!pt.isUpdateFinish()) return 4;
    }
    std::puts("ok");
    return 0;
}
''')
    exe = tmp_path / "stub"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I", ROOT, "-o", str(exe), str(stub)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=30)
    assert out.returncode == 0 and out.stdout.strip() == "ok", (out.returncode, out.stdout, out.stderr)


def test_node_create_without_rccl_reports_it_and_does_not_crash(cornell):
    """RT_RCCL_LIB names the only library the loader tries: with a path that does not exist rt_node_create must come
    back with RT_ERR_NO_DEVICE and the loader's message (it is loaded before any device is touched), not crash."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys
        os.environ["RT_RCCL_LIB"] = "/nonexistent/librccl-missing.so"
        sys.path.insert(0, %r)
        from raytracer_amd import Node, api, scenes
        try:
            Node(scenes.cornell_box(), 64, 64, [0])
        except api.RtError as e:
            print("code", e.code, str(e))
            sys.exit(0)
        sys.exit(1)
    """ % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "code -3" in r.stdout and "cannot load RCCL (/nonexistent/librccl-missing.so)" in r.stdout


def test_demo_host_rejects_a_malformed_device_list(tmp_path):
    import subprocess
    demo = os.path.join(ROOT, "raytracer_amd", "host", "demo_host")
    if not os.path.exists(demo):
        pytest.skip("demo_host not built")
    # a readable scene file is needed to get as far as the device list: header of a zero-triangle scene is refused
    # later, the parser comes first
    scene = tmp_path / "scene.bin"
    import numpy as np
    np.zeros(16, np.uint32).tofile(scene)
    for bad in ("node", "0;1", "", "1,,2", "-1"):
        r = subprocess.run([demo, str(scene), "8", "8", "1", str(tmp_path / "f.f32"), bad], capture_output=True, text=True, timeout=60)
        assert r.returncode == 2 and "device list" in r.stderr, (bad, r.returncode, r.stderr)
