"""Procedural, seeded scenes and ray sets.

The reference ships no assets (its app loads ``models/lambo/lambo`` and an HDR
map that are not in the tree, src/app/App.cpp:30,37), so every workload here is
generated.  A scene is the flat view ``BVH::runLBVH`` has of it
(src/scene/BVH.cpp:293-317): concatenated vertices, already-offset indices, a
model id per triangle, one material per model, plus the light list of
src/app/App.cpp:34-38 (a point light and an environment map).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

LIGHT_POINT = 0  # src/lights/LightSource.hpp:12-17
LIGHT_AREA = 2
LIGHT_ENV = 3

GLOSSY_NONE, GLOSSY_BECKMANN, GLOSSY_GGX = 0, 1, 2


@dataclass
class Light:
    type: int
    pos: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    intensity: np.ndarray = field(default_factory=lambda: np.ones(3, np.float32))
    env: Optional[np.ndarray] = None  # (H, W, 4) float32
    tri: Optional[np.ndarray] = None  # area: (3, 3) float32 corners; emits `intensity` on the side of cross(p1-p0, p2-p0)


@dataclass
class Camera:
    pos: np.ndarray
    right: np.ndarray
    up: np.ndarray
    forward: np.ndarray

    def packed(self) -> np.ndarray:
        return np.concatenate([self.pos, self.right, self.up, self.forward]).astype(np.float32)


@dataclass
class Scene:
    pos: np.ndarray        # (V, 3) f32
    normal: np.ndarray     # (V, 3)
    uv: np.ndarray         # (V, 2)
    tangent: np.ndarray    # (V, 3)
    binormal: np.ndarray   # (V, 3)
    indices: np.ndarray    # (3T,) u32, already offset
    tri_model: np.ndarray  # (T,) i32
    materials: np.ndarray  # (M, 4) f32: rgb diffuseColor + roughness
    lights: List[Light]
    camera: Camera
    name: str = ""
    textures: List[np.ndarray] = field(default_factory=list)   # each (H, W, 4) u8, level 0 only
    material_texture: Optional[np.ndarray] = None              # (M,) i32 texture index per model, -1 = none
    material_glossy: Optional[np.ndarray] = None               # (M,) i32: 0 none, 1 Beckmann, 2 GGX lobe

    def __post_init__(self):
        if self.material_glossy is None:
            self.material_glossy = np.zeros(self.materials.shape[0], np.int32)
        self.material_glossy = np.ascontiguousarray(self.material_glossy, np.int32)
        if self.material_texture is None:
            self.material_texture = np.full(self.materials.shape[0], -1, np.int32)
        self.material_texture = np.ascontiguousarray(self.material_texture, np.int32)
        self.textures = [np.ascontiguousarray(t, np.uint8) for t in self.textures]

    @property
    def n_triangles(self) -> int:
        return int(self.indices.size // 3)

    @property
    def n_models(self) -> int:
        return int(self.materials.shape[0])


def make_camera(pos, forward, world_up=(0.0, 1.0, 0.0)) -> Camera:
    """Camera vectors as Integrator::nextFrame snapshots them
    (src/cpu/Integrator.cpp:85-92): up = normalize(cross(right, forward))."""
    f = np.asarray(forward, np.float64)
    f = f / np.linalg.norm(f)
    r = np.cross(f, np.asarray(world_up, np.float64))
    r = r / np.linalg.norm(r)
    f32 = f.astype(np.float32)
    r32 = r.astype(np.float32)
    c = np.cross(r32, f32).astype(np.float32)
    up = (c * np.float32(1.0 / np.sqrt(np.float32(np.dot(c, c))))).astype(np.float32)
    return Camera(np.asarray(pos, np.float32), r32, up, f32)


class _Builder:
    def __init__(self):
        self.pos, self.nrm, self.uv, self.tan, self.bin = [], [], [], [], []
        self.idx, self.tri_model, self.materials = [], [], []
        self.nv = 0
        self.model = -1

    def begin_model(self, color, roughness=0.5):
        self.materials.append([color[0], color[1], color[2], roughness])
        self.model += 1

    def add_grid(self, P, uv=None, flip=False):
        """P: (ny, nx, 3) grid of points -> 2*(ny-1)*(nx-1) triangles with
        smooth normals and a tangent frame from the grid derivatives."""
        P = np.asarray(P, np.float64)
        ny, nx, _ = P.shape
        du = np.gradient(P, axis=1)
        dv = np.gradient(P, axis=0)
        n = np.cross(du, dv)
        if flip:
            n = -n
        ln = np.linalg.norm(n, axis=-1, keepdims=True)
        n = np.where(ln > 1e-20, n / np.maximum(ln, 1e-20), np.array([0.0, 1.0, 0.0]))
        t = du - n * np.sum(du * n, axis=-1, keepdims=True)
        lt = np.linalg.norm(t, axis=-1, keepdims=True)
        alt = np.cross(n, np.array([0.0, 0.0, 1.0]))
        alt_l = np.linalg.norm(alt, axis=-1, keepdims=True)
        alt = np.where(alt_l > 1e-6, alt / np.maximum(alt_l, 1e-20), np.cross(n, np.array([1.0, 0.0, 0.0])))
        t = np.where(lt > 1e-12, t / np.maximum(lt, 1e-20), alt)
        b = np.cross(n, t)
        if uv is None:
            uu, vv = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1, ny))
            uv = np.stack([uu, vv], -1)
        ii, jj = np.meshgrid(np.arange(nx - 1), np.arange(ny - 1))
        a = (jj * nx + ii).ravel()
        q = np.stack([a, a + 1, a + nx + 1, a, a + nx + 1, a + nx], -1)
        if flip:
            q = q[:, [0, 2, 1, 3, 5, 4]]
        self.idx.append((q.ravel() + self.nv).astype(np.uint32))
        self.tri_model.append(np.full(q.shape[0] * 2, self.model, np.int32))
        self.pos.append(P.reshape(-1, 3))
        self.nrm.append(n.reshape(-1, 3))
        self.uv.append(np.asarray(uv, np.float64).reshape(-1, 2))
        self.tan.append(t.reshape(-1, 3))
        self.bin.append(b.reshape(-1, 3))
        self.nv += ny * nx

    def add_quad(self, p0, p1, p2, p3, nx=2, ny=2, flip=False):
        """Bilinear patch p0->p1 (u) and p0->p3 (v)."""
        p0, p1, p2, p3 = (np.asarray(p, np.float64) for p in (p0, p1, p2, p3))
        u = np.linspace(0, 1, nx)[None, :, None]
        v = np.linspace(0, 1, ny)[:, None, None]
        P = (1 - u) * (1 - v) * p0 + u * (1 - v) * p1 + u * v * p2 + (1 - u) * v * p3
        self.add_grid(P, flip=flip)

    def add_box(self, lo, hi, n=2, rot_y=0.0):
        lo = np.asarray(lo, np.float64)
        hi = np.asarray(hi, np.float64)
        c = (lo + hi) / 2
        cs, sn = np.cos(rot_y), np.sin(rot_y)

        def R(p):
            p = np.asarray(p, np.float64) - c
            return np.array([cs * p[0] + sn * p[2], p[1], -sn * p[0] + cs * p[2]]) + c

        x0, y0, z0 = lo
        x1, y1, z1 = hi
        faces = [
            ((x0, y1, z1), (x1, y1, z1), (x1, y1, z0), (x0, y1, z0)),  # top
            ((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)),  # bottom
            ((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)),  # front
            ((x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0)),  # back
            ((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)),  # left
            ((x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1)),  # right
        ]
        for f in faces:
            self.add_quad(*[R(p) for p in f], nx=n, ny=n)

    def add_lathe(self, center, radius_fn, y0, y1, nseg, nring):
        """Surface of revolution around the y axis through `center`."""
        th = np.linspace(0, 2 * np.pi, nseg + 1)[None, :]
        yy = np.linspace(y0, y1, nring + 1)[:, None]
        r = radius_fn((yy - y0) / (y1 - y0))
        P = np.stack([center[0] + r * np.cos(th), yy + 0 * th + center[1], center[2] - r * np.sin(th)], -1)
        self.add_grid(P)

    def finish(self, lights, camera, name, jitter=0.0, seed=1) -> Scene:
        pos = np.concatenate(self.pos)
        if jitter > 0:
            rng = np.random.default_rng(seed)
            pos = pos + rng.uniform(-jitter, jitter, pos.shape)
        return Scene(
            pos=np.ascontiguousarray(pos, np.float32),
            normal=np.ascontiguousarray(np.concatenate(self.nrm), np.float32),
            uv=np.ascontiguousarray(np.concatenate(self.uv), np.float32),
            tangent=np.ascontiguousarray(np.concatenate(self.tan), np.float32),
            binormal=np.ascontiguousarray(np.concatenate(self.bin), np.float32),
            indices=np.ascontiguousarray(np.concatenate(self.idx), np.uint32),
            tri_model=np.ascontiguousarray(np.concatenate(self.tri_model), np.int32),
            materials=np.ascontiguousarray(np.asarray(self.materials), np.float32),
            lights=lights, camera=camera, name=name)


def sky_env(width=256, height=128, seed=7) -> np.ndarray:
    """Procedural sky: horizon-to-zenith gradient, a warm sun lobe, dim ground."""
    v = (np.arange(height) + 0.5) / height
    u = (np.arange(width) + 0.5) / width
    uu, vv = np.meshgrid(u, v)
    up = np.cos(vv * np.pi)
    sky = np.stack([0.35 + 0.25 * (1 - up), 0.55 + 0.2 * (1 - up), 0.95 + 0 * up], -1) * np.clip(up + 0.15, 0.05, 1)[..., None]
    ground = np.array([0.12, 0.10, 0.08]) * np.ones_like(sky)
    img = np.where((up > 0)[..., None], sky, ground)
    sun = np.exp(-(((uu - 0.3) * 2 * np.pi) ** 2 + ((vv - 0.22) * np.pi) ** 2) / 0.004)
    img = img + sun[..., None] * np.array([40.0, 36.0, 28.0])
    out = np.ones((height, width, 4), np.float32)
    out[..., :3] = img.astype(np.float32)
    return out


def cornell_box(env: bool = False) -> Scene:
    """Cornell box: 5 walls (10 triangles) + 2 closed boxes (24 triangles) in 7
    models, one point light just below the ceiling centre (the reference has
    no area lights), camera at (0, 1, 3.4) with unit right/forward."""
    b = _Builder()
    white, red, green = (0.73, 0.73, 0.73), (0.65, 0.05, 0.05), (0.12, 0.45, 0.15)
    b.begin_model(white); b.add_quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1))          # floor
    b.begin_model(white); b.add_quad((-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1))          # ceiling
    b.begin_model(white); b.add_quad((-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1))        # back
    b.begin_model(red);   b.add_quad((-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1))        # left
    b.begin_model(green); b.add_quad((1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1))            # right
    b.begin_model(white); b.add_box((0.15, 0.0, 0.05), (0.75, 0.6, 0.65), rot_y=-0.3)         # short box
    b.begin_model(white); b.add_box((-0.75, 0.0, -0.7), (-0.15, 1.2, -0.1), rot_y=0.3)        # tall box
    lights = [Light(LIGHT_POINT, np.array([0.0, 1.9, 0.0], np.float32), np.array([1.5, 1.5, 1.5], np.float32))]
    if env:
        lights.append(Light(LIGHT_ENV, intensity=np.array([1.0, 1.0, 1.0], np.float32), env=sky_env(64, 32)))
    cam = Camera(np.array([0, 1, 3.4], np.float32), np.array([1, 0, 0], np.float32),
                 np.array([0, 1, 0], np.float32), np.array([0, 0, -1], np.float32))
    return b.finish(lights, cam, "cornell")


def _atrium(b: _Builder, origin=(0.0, 0.0, 0.0), detail=1.0, palette_shift=0):
    """One Sponza-like atrium (about 37 x 15 x 23 m): floor, outer walls,
    two storeys of colonnades with arches, gallery slabs, drapes, a roof with
    an opening, vases and spheres.  `detail` scales the tessellation."""
    ox, oy, oz = origin

    def T(P):
        return np.asarray(P, np.float64) + np.array([ox, oy, oz])

    def n_(x):
        return max(2, int(round(x * detail)))

    stone = [(0.62, 0.58, 0.52), (0.55, 0.5, 0.45), (0.7, 0.66, 0.6)]
    s = lambda i: stone[(i + palette_shift) % 3]
    X, Y, Z = 18.5, 15.0, 11.5

    def patch(p0, p1, p3, nx, ny, bump=0.0, flip=False):
        p0, p1, p3 = (np.asarray(p, np.float64) for p in (p0, p1, p3))
        u = np.linspace(0, 1, nx)[None, :, None]
        v = np.linspace(0, 1, ny)[:, None, None]
        P = p0 + u * (p1 - p0) + v * (p3 - p0)
        if bump:
            nrm = np.cross(p1 - p0, p3 - p0)
            nrm /= np.linalg.norm(nrm)
            P = P + nrm * bump * (np.sin(u * 37.0) * np.cos(v * 29.0))
        b.add_grid(T(P), flip=flip)

    # floor
    b.begin_model(s(0), 0.6)
    patch((-X, 0, Z), (X, 0, Z), (-X, 0, -Z), n_(161), n_(101), bump=0.01)
    # outer walls
    b.begin_model(s(1), 0.7)
    patch((-X, 0, -Z), (X, 0, -Z), (-X, Y, -Z), n_(129), n_(49), bump=0.02)
    patch((X, 0, Z), (-X, 0, Z), (X, Y, Z), n_(129), n_(49), bump=0.02)
    b.begin_model(s(2), 0.7)
    patch((-X, 0, Z), (-X, 0, -Z), (-X, Y, Z), n_(81), n_(49), bump=0.02)
    patch((X, 0, -Z), (X, 0, Z), (X, Y, -Z), n_(81), n_(49), bump=0.02)
    # roof ring with a central opening
    b.begin_model(s(0), 0.8)
    patch((-X, Y, -Z), (X, Y, -Z), (-X, Y, -4), n_(97), n_(21))
    patch((-X, Y, 4), (X, Y, 4), (-X, Y, Z), n_(97), n_(21))
    patch((-X, Y, -4), (-9, Y, -4), (-X, Y, 4), n_(25), n_(21))
    patch((9, Y, -4), (X, Y, -4), (9, Y, 4), n_(25), n_(21))
    # gallery slabs (first-storey ceiling) on both long sides and both ends
    b.begin_model(s(1), 0.5)
    for z0, z1 in ((-Z, -6.0), (6.0, Z)):
        patch((-X, 6.8, z1), (X, 6.8, z1), (-X, 6.8, z0), n_(97), n_(13))
        patch((-X, 7.2, z0), (X, 7.2, z0), (-X, 7.2, z1), n_(97), n_(13))
        zi = z1 if z0 < 0 else z0
        patch((-X, 6.8, zi), (X, 6.8, zi), (-X, 7.2, zi), n_(97), 2, flip=(z0 > 0))
    # columns: two storeys, two rows
    col_x = np.linspace(-15.0, 15.0, 11)
    mi = 0
    for storey, (y0, y1, rad) in enumerate(((0.0, 5.0, 0.45), (7.2, 11.5, 0.35))):
        for zrow in (-6.0, 6.0):
            b.begin_model(s(mi), 0.4); mi += 1
            for cx in col_x:
                prof = lambda t, rad=rad: rad * (1.0 + 0.25 * np.exp(-((t - 0.03) / 0.04) ** 2)
                                                 + 0.3 * np.exp(-((t - 0.97) / 0.04) ** 2) + 0.03 * np.cos(t * 0 + 0))
                b.add_lathe((cx + ox, oy, zrow + oz), prof, y0, y1, n_(24), n_(14))
    # arches between neighbouring columns (half rings), both storeys
    for storey, (ybase, r_in) in enumerate(((5.0, 1.05), (11.5, 1.15))):
        b.begin_model(s(storey + 1), 0.5)
        for zrow in (-6.0, 6.0):
            for i in range(len(col_x) - 1):
                cx = 0.5 * (col_x[i] + col_x[i + 1])
                half = 0.5 * (col_x[i + 1] - col_x[i]) - 0.1
                a = np.linspace(0, np.pi, n_(25))[None, :]
                w = np.linspace(-0.35, 0.35, n_(5))[:, None]
                P = np.stack([cx - half * np.cos(a) + 0 * w, ybase + r_in * 1.6 * np.sin(a) + 0 * w, zrow + w + 0 * a], -1)
                b.add_grid(T(P))
    # drapes: wavy hanging sheets between the upper columns
    cols = [(0.7, 0.1, 0.1), (0.1, 0.25, 0.6), (0.15, 0.5, 0.2), (0.75, 0.6, 0.15)]
    for k in range(8):
        b.begin_model(cols[(k + palette_shift) % 4], 0.9)
        x0 = -14.0 + k * 4.0
        zrow = -5.2 if k % 2 == 0 else 5.2
        u = np.linspace(0, 1, n_(49))[None, :]
        v = np.linspace(0, 1, n_(65))[:, None]
        P = np.stack([x0 + 2.6 * u + 0 * v,
                      11.0 - 5.5 * v + 0.12 * np.sin(u * 18.0) * v,
                      zrow + 0.25 * np.sin(u * 25.0 + k) * (0.3 + v) + 0.1 * np.cos(v * 9.0)], -1)
        b.add_grid(T(P))
    # vases along the centre line
    b.begin_model((0.35, 0.55, 0.6), 0.3)
    for k in range(10):
        cx = -13.5 + 3.0 * k
        prof = lambda t: 0.18 + 0.35 * np.sin(np.pi * np.clip(t, 0, 1)) ** 2 * (1.2 - t)
        b.add_lathe((cx + ox, oy, (1.8 if k % 2 else -1.8) + oz), prof, 0.0, 1.3, n_(32), n_(24))
    # ornament spheres on the gallery
    b.begin_model((0.8, 0.75, 0.3), 0.2)
    for k in range(8):
        cx = -14.0 + 4.0 * k
        prof = lambda t: 0.5 * np.sqrt(np.clip(1 - (2 * t - 1) ** 2, 0, 1))
        b.add_lathe((cx + ox, oy, (7.5 if k % 2 else -7.5) + oz), prof, 7.2, 8.2, n_(48), n_(32))


def sponza_class(detail: float = 1.1, env_size=(2048, 1024)) -> Scene:
    """Procedural atrium standing in for Sponza (no asset offline): about
    262 k triangles at the default detail in ~20 models, AABB about 37 x 15 x 23 m,
    vertices jittered so Morton codes rarely collide; a point light and a
    procedural sky (src/app/App.cpp:34-38)."""
    b = _Builder()
    _atrium(b, detail=detail)
    lights = [
        Light(LIGHT_POINT, np.array([0.0, 10.0, 0.0], np.float32), np.array([100.0, 100.0, 100.0], np.float32)),
        Light(LIGHT_ENV, intensity=np.array([10.0, 10.0, 10.0], np.float32), env=sky_env(*env_size)),
    ]
    cam = make_camera((-16.5, 3.2, 1.3), (1.0, 0.08, -0.06))
    return b.finish(lights, cam, f"sponza_class(detail={detail})", jitter=2e-4, seed=1)


def bistro_class(blocks=(4, 3), detail: float = 1.1, n_point_lights: int = 64, env_size=(1024, 512), area_lights: bool = False) -> Scene:
    """A street block of atria: >= 2.8 M triangles at blocks=(4,3), many point
    lights (the reference has no area lights).  area_lights=True puts a small
    downward-facing emissive triangle in place of each point light (the
    build's own light kind, behind the same sampleLi / pdfLi / le interface)."""
    b = _Builder()
    k = 0
    for bx in range(blocks[0]):
        for bz in range(blocks[1]):
            _atrium(b, origin=(bx * 40.0, 0.0, bz * 26.0), detail=detail, palette_shift=k)
            k += 1
    rng = np.random.default_rng(3)
    lights = []
    for i in range(n_point_lights):
        p = np.array([rng.uniform(-15, blocks[0] * 40 - 25), rng.uniform(3, 12), rng.uniform(-9, blocks[1] * 26 - 17)], np.float32)
        if area_lights:
            # a 0.6 m triangle facing down (normal = cross(p1 - p0, p2 - p0) = -y), about the point light's power
            tri = np.stack([p + np.array(d, np.float32) for d in ((-0.3, 0.0, -0.2), (0.3, 0.0, -0.2), (0.0, 0.0, 0.4))])
            lights.append(Light(LIGHT_AREA, intensity=np.full(3, 160.0, np.float32), tri=tri))
        else:
            lights.append(Light(LIGHT_POINT, p, np.full(3, 30.0, np.float32)))
    lights.append(Light(LIGHT_ENV, intensity=np.array([5.0, 5.0, 5.0], np.float32), env=sky_env(*env_size)))
    cam = make_camera((-16.5, 3.2, 1.3), (1.0, 0.05, 0.35))
    return b.finish(lights, cam, f"bistro_class(blocks={blocks},detail={detail},area_lights={area_lights})", jitter=2e-4, seed=2)


# --------------------------------------------------------------- textures
def checker_texture(size=64, cells=8, a=(235, 235, 235), b=(40, 40, 40)) -> np.ndarray:
    y, x = np.mgrid[0:size, 0:size]
    m = (((x * cells) // size + (y * cells) // size) & 1).astype(bool)
    img = np.where(m[..., None], np.array(a, np.uint8), np.array(b, np.uint8)).astype(np.uint8)
    return np.concatenate([img, np.full((size, size, 1), 255, np.uint8)], -1)


def brick_texture(width=128, height=64, seed=11) -> np.ndarray:
    """Running-bond bricks with mortar lines and per-brick tint; not square on purpose."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width]
    row = y // 8
    xs = (x + (row & 1) * 8) % width
    col = xs // 16
    mortar = ((y % 8) == 0) | ((xs % 16) == 0)
    tint = rng.integers(-25, 25, (height // 8 + 1, width // 16 + 1))
    base = np.stack([170 + tint[row, col], 80 + tint[row, col] // 2, 60 + tint[row, col] // 3], -1)
    img = np.where(mortar[..., None], np.array([200, 196, 188]), base)
    noise = rng.integers(-8, 9, (height, width, 1))
    img = np.clip(img + noise, 0, 255).astype(np.uint8)
    return np.concatenate([img, np.full((height, width, 1), 255, np.uint8)], -1)


def noise_texture(size=256, seed=5) -> np.ndarray:
    """Sum of a few octaves of value noise in three decorrelated channels."""
    rng = np.random.default_rng(seed)
    img = np.zeros((size, size, 3))
    for o, amp in ((4, 0.5), (16, 0.3), (64, 0.2)):
        g = rng.uniform(0, 1, (o, o, 3))
        img += amp * np.kron(g, np.ones((size // o, size // o, 1)))
    img = np.clip(40 + img * 215, 0, 255).astype(np.uint8)
    return np.concatenate([img, np.full((size, size, 1), 255, np.uint8)], -1)


def with_textures(scene: Scene, uv_scale: float = 1.0, every: int = 1) -> Scene:
    """The same geometry with procedural diffuse maps bound to every `every`-th
    model (checker, bricks, noise in turn) and UVs scaled so that they repeat."""
    tex = [checker_texture(), brick_texture(), noise_texture()]
    mt = np.full(scene.n_models, -1, np.int32)
    k = 0
    for m in range(0, scene.n_models, every):
        mt[m] = k % len(tex)
        k += 1
    s = Scene(**{f: getattr(scene, f) for f in ("pos", "normal", "tangent", "binormal", "indices", "tri_model",
                                                "materials", "lights", "camera")},
              uv=np.ascontiguousarray(scene.uv * np.float32(uv_scale), np.float32),
              name=scene.name + "+tex", textures=tex, material_texture=mt)
    return s


def replace(scene: Scene, **kw) -> Scene:
    """A copy of the scene with some fields replaced."""
    f = {k: getattr(scene, k) for k in ("pos", "normal", "uv", "tangent", "binormal", "indices", "tri_model", "materials",
                                        "lights", "camera", "name", "textures", "material_texture", "material_glossy")}
    f.update(kw)
    return Scene(**f)


def with_glossy(scene: Scene, kind: int = GLOSSY_GGX, every: int = 1) -> Scene:
    """The same scene with the microfacet lobe on every `every`-th model."""
    mg = np.zeros(scene.n_models, np.int32)
    mg[::every] = kind
    return replace(scene, material_glossy=mg, name=scene.name + f"+glossy{kind}")


def area_quad(p0, p1, p2, p3, radiance) -> List[Light]:
    """Two emissive triangles (p0,p1,p2), (p0,p2,p3) emitting on the side of cross(p1-p0, p2-p0)."""
    pts = [np.asarray(p, np.float32) for p in (p0, p1, p2, p3)]
    rad = np.asarray(radiance, np.float32)
    return [Light(LIGHT_AREA, intensity=rad, tri=np.stack([pts[0], pts[1], pts[2]])),
            Light(LIGHT_AREA, intensity=rad, tri=np.stack([pts[0], pts[2], pts[3]]))]


def cornell_area(glossy: int = GLOSSY_NONE) -> Scene:
    """Cornell box lit by a ceiling panel (two area lights just below the
    ceiling, facing down) instead of the point light."""
    s = cornell_box()
    panel = area_quad((-0.3, 1.98, -0.3), (0.3, 1.98, -0.3), (0.3, 1.98, 0.3), (-0.3, 1.98, 0.3), (18.0, 16.0, 12.0))
    s = replace(s, lights=panel, name="cornell_area")
    return with_glossy(s, glossy) if glossy else s


# --------------------------------------------------------------- ray sets
def _pcg32(n: int, seed: int) -> np.ndarray:
    """n 32-bit outputs of pcg32 (XSH-RR), vectorised over a jump-ahead-free
    stream split: output i comes from state advanced i times."""
    mult = np.uint64(6364136223846793005)
    inc = np.uint64((seed << 1) | 1)
    # advance sequentially in blocks; numpy uint64 arithmetic wraps
    out = np.empty(n, np.uint32)
    state = np.uint64(seed) + inc
    with np.errstate(over="ignore"):
        # LCG jump: state_i = a^i * s0 + c*(a^i-1)/(a-1); compute a^i and the
        # geometric sums by cumulative products over blocks
        a_pow = np.empty(n, np.uint64)
        c_sum = np.empty(n, np.uint64)
        a, c = np.uint64(1), np.uint64(0)
        # doubling construction
        a_pow[0], c_sum[0] = a, c
        filled = 1
        cur_a, cur_c = mult, inc
        while filled < n:
            m = min(filled, n - filled)
            a_pow[filled:filled + m] = a_pow[:m] * cur_a
            c_sum[filled:filled + m] = c_sum[:m] * cur_a + cur_c
            filled += m
            cur_c = cur_c * cur_a + cur_c
            cur_a = cur_a * cur_a
        st = a_pow * state + c_sum
        xorshifted = (((st >> np.uint64(18)) ^ st) >> np.uint64(27)).astype(np.uint32)
        rot = (st >> np.uint64(59)).astype(np.uint32)
        out = (xorshifted >> rot) | (xorshifted << ((np.uint32(32) - rot) & np.uint32(31)))
    return out


def _u01(bits: np.ndarray) -> np.ndarray:
    return ((bits >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def incoherent_rays(scene: Scene, n: int, seed: int = 0x5EED) -> np.ndarray:
    """(6, n) f32 SoA: origins uniform in the scene AABB shrunk by 10 %,
    directions uniform on the sphere."""
    bits = _pcg32(5 * n, seed)
    u = _u01(bits).reshape(5, n)
    lo, hi = scene.pos.min(0), scene.pos.max(0)
    c, h = (lo + hi) / 2, (hi - lo) / 2 * 0.9
    org = (c[:, None] + h[:, None] * (2 * u[:3] - 1)).astype(np.float32)
    z = 1 - 2 * u[3]
    r = np.sqrt(np.maximum(0, 1 - z * z))
    phi = 2 * np.pi * u[4]
    d = np.stack([r * np.cos(phi), r * np.sin(phi), z]).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([org, d], 0), np.float32)


def bounce_rays(scene: Scene, W: int, H: int, n: int, seed: int = 0x5EED, device: int = 0) -> np.ndarray:
    """(6, n) f32 SoA: the rays a path tracer sends after the primary hit
    (origins on the surfaces the camera sees, directions sampled from the
    BSDF), taken from the tracer itself: depth-2 passes at W x H are rendered
    until n rays have been collected."""
    from .api import Context
    ctx = Context(scene, W, H, max_depth=2, seed=seed, device=device, batch=1)
    ctx.set_camera(scene.camera)
    out, have = np.empty((6, n), np.float32), 0
    for _ in range(256):
        ctx.render(1)
        r = ctx.last_bounce_rays()
        k = min(n - have, r.shape[1])
        out[:, have:have + k] = r[:, :k]
        have += k
        if have >= n:
            break
    ctx.close()
    if have < n:
        raise RuntimeError("the scene yields too few bounce rays")
    return out


def camera_rays_numpy(cam: Camera, W: int, H: int) -> np.ndarray:
    """(6, W*H) f32 SoA of the pinhole rays of src/cpu/PathTracer.cpp:43-52,
    row-major, evaluated in float32 with the same operation order."""
    f = np.float32
    x = np.arange(W, dtype=np.float32)[None, :]
    y = np.arange(H, dtype=np.float32)[:, None]
    ni = (x / f(W)) - f(0.5) + 0 * y
    nj = (y / f(H)) - f(0.5) + 0 * x
    nj = -nj
    nj = nj * (f(H) / f(W))
    ip = [ni * cam.right[k] + nj * cam.up[k] + cam.forward[k] for k in range(3)]
    d2 = ip[0] * ip[0] + ip[1] * ip[1] + ip[2] * ip[2]
    inv = f(1.0) / np.sqrt(d2)
    d = [(c * inv).astype(np.float32).ravel() for c in ip]
    n = W * H
    o = [np.full(n, cam.pos[k], np.float32) for k in range(3)]
    return np.ascontiguousarray(np.stack(o + d), np.float32)
