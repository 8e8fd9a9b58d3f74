"""Seeded synthetic scenes: the stand-in for the reference's scene loader.

The glTF models BASELINE.json names (SciFiHelmet, FlightHelmet, Sponza) are not
available offline, so every config is restated on procedural geometry and
value-noise textures.  This module plays the
role of App::load_scene (reference src/app.cpp:173-385): it produces exactly
what that function hands to the Renderer -- RGBA8 texture triples, Vertex /
index arrays, an object list and the Scene defaults of src/app.hpp:42-62 --
and lives ABOVE the C-ABI; nothing here is on the hot path.

Master seed 0x41524354 ("ARCT"); all randomness comes from numpy's PCG64, which
is bit-reproducible across machines, so the CPU oracle and the HIP path see
identical bytes.
"""
import colorsys

import numpy as np

from .scene import (LIGHT_DTYPE, VERTEX_DTYPE, SceneDesc, make_lights, make_objects, rotation_y, scaling,
                    translation, TM_ACES, TM_REINHARD)

SEED = 0x41524354


# ----------------------------------------------------------------------------
# textures
# ----------------------------------------------------------------------------
def value_noise(rng, size, cells):
    """tileable smooth value noise in [0,1], shape (size,size)."""
    g = rng.random((cells, cells), dtype=np.float32)
    t = (np.arange(size, dtype=np.float32) + 0.5) * (cells / size)
    i0 = np.floor(t).astype(np.int64) % cells
    i1 = (i0 + 1) % cells
    f = t - np.floor(t)
    f = f * f * (3.0 - 2.0 * f)
    rows = g[i0][:, None, :] * (1 - f)[:, None, None] + g[i1][:, None, :] * f[:, None, None]   # (size,1,cells)
    rows = rows[:, 0, :]
    return rows[:, i0] * (1 - f)[None, :] + rows[:, i1] * f[None, :]


def fractal_noise(rng, size, cells, octaves=3):
    out, amp, tot = np.zeros((size, size), np.float32), 1.0, 0.0
    for o in range(octaves):
        c = min(cells << o, size)
        out += amp * value_noise(rng, size, c)
        tot += amp
        amp *= 0.5
    return out / tot


def make_material_textures(rng, size, hue=None):
    """(diffuse sRGB, normal, metal-rough) RGBA8 arrays per SURVEY 8(d):
    diffuse bytes in [30,230]; normal (128,128,255) +- 40 in x/y; rough (G) in
    [13,255]; metal (B) in {0,255} with P(metal) = 0.2; R/A = 255."""
    cells = max(4, size // 64)
    hue = rng.random() if hue is None else hue
    base = np.array(colorsys.hsv_to_rgb(hue, 0.25 + 0.5 * rng.random(), 0.9), np.float32)
    lum = fractal_noise(rng, size, cells)
    tint = np.stack([fractal_noise(rng, size, cells, 2) for _ in range(3)], -1)
    d = 30.0 + 200.0 * np.clip(lum[..., None] * (0.55 + 0.45 * base) * (0.8 + 0.4 * tint), 0.0, 1.0)
    diffuse = np.empty((size, size, 4), np.uint8)
    diffuse[..., :3] = np.rint(d).astype(np.uint8)
    diffuse[..., 3] = 255
    normal = np.empty((size, size, 4), np.uint8)
    normal[..., 0] = np.rint(128.0 + 80.0 * (fractal_noise(rng, size, cells * 2) - 0.5)).astype(np.uint8)
    normal[..., 1] = np.rint(128.0 + 80.0 * (fractal_noise(rng, size, cells * 2) - 0.5)).astype(np.uint8)
    normal[..., 2] = 255
    normal[..., 3] = 255
    mr = np.empty((size, size, 4), np.uint8)
    mr[..., 0] = 255
    mr[..., 1] = np.rint(13.0 + 242.0 * fractal_noise(rng, size, cells)).astype(np.uint8)
    m = value_noise(rng, size, cells)
    mr[..., 2] = np.where(m > np.quantile(m, 0.8), 255, 0).astype(np.uint8)
    mr[..., 3] = 255
    return diffuse, normal, mr


def fallback_textures():
    """assets/white.png and assets/normal.png decoded (16x16): the loader's
    fallbacks (src/app.cpp:194-245) => base 1, metal 1, rough 1, flat normal."""
    white = np.full((16, 16, 4), 255, np.uint8)
    normal = np.empty((16, 16, 4), np.uint8)
    normal[...] = (128, 128, 255, 255)
    return white, normal, white.copy()


# ----------------------------------------------------------------------------
# meshes: parametric grids with analytic normal / tangent (dP/du) / bitangent (dP/dv)
# ----------------------------------------------------------------------------
def _unit(a):
    n = np.linalg.norm(a, axis=-1, keepdims=True)
    return a / np.where(n == 0, 1, n)


def grid_mesh(P, N, T, B, UV):
    """arrays (nv+1, nu+1, k) -> (vertices, indices); triangles wound counter-clockwise
    seen from the side N points to (front face, forward_pass.cpp:143-144)."""
    nv, nu = P.shape[0] - 1, P.shape[1] - 1
    v = np.zeros((nv + 1) * (nu + 1), VERTEX_DTYPE)
    v["position"], v["normal"] = P.reshape(-1, 3), N.reshape(-1, 3)
    v["tangent"], v["bitangent"], v["tex_coords"] = T.reshape(-1, 3), B.reshape(-1, 3), UV.reshape(-1, 2)
    j, i = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = (j * (nu + 1) + i).ravel()
    b, c, d = a + 1, a + nu + 2, a + nu + 1
    # orientation: does (P_u x P_v) point along N ?
    pu, pv = P[0, 1] - P[0, 0], P[1, 0] - P[0, 0]
    k = nv // 2
    pu, pv = P[k, nu // 2 + 1] - P[k, nu // 2], P[k + 1, nu // 2] - P[k, nu // 2]
    flip = np.dot(np.cross(pu, pv), N[k, nu // 2]) < 0
    tris = np.stack([a, b, c, a, c, d], 1) if not flip else np.stack([a, c, b, a, d, c], 1)
    return v, tris.reshape(-1).astype(np.uint32)


def _uvgrid(nu, nv):
    u, v = np.meshgrid(np.linspace(0, 1, nu + 1, dtype=np.float32), np.linspace(0, 1, nv + 1, dtype=np.float32))
    return u, v


def uv_sphere(r=1.0, nu=128, nv=64, uv_scale=(1.0, 1.0)):
    u, v = _uvgrid(nu, nv)
    th, ph = 2 * np.pi * u, np.pi * (0.001 + 0.998 * v)   # keep off the poles: no zero-area fans
    n = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], -1).astype(np.float32)
    t = np.stack([-np.sin(th), np.zeros_like(th), np.cos(th)], -1).astype(np.float32)
    b = np.stack([np.cos(ph) * np.cos(th), -np.sin(ph), np.cos(ph) * np.sin(th)], -1).astype(np.float32)
    return grid_mesh(r * n, n, t, b, np.stack([u * uv_scale[0], v * uv_scale[1]], -1))


def torus(R=1.0, r=0.35, nu=96, nv=48):
    u, v = _uvgrid(nu, nv)
    th, ph = 2 * np.pi * u, 2 * np.pi * v
    cx, cz = np.cos(th), np.sin(th)
    n = np.stack([np.cos(ph) * cx, np.sin(ph), np.cos(ph) * cz], -1).astype(np.float32)
    p = np.stack([(R + r * np.cos(ph)) * cx, r * np.sin(ph), (R + r * np.cos(ph)) * cz], -1).astype(np.float32)
    t = np.stack([-cz, np.zeros_like(cz), cx], -1).astype(np.float32)
    b = np.stack([-np.sin(ph) * cx, np.cos(ph), -np.sin(ph) * cz], -1).astype(np.float32)
    return grid_mesh(p, n, t, b, np.stack([u * 4, v * 2], -1))


def cylinder(r=0.35, y0=0.0, y1=6.0, nu=32, nv=64, uv_scale=(2.0, 6.0), inward=False):
    """open tube around the Y axis; inward=True makes the inside the front face."""
    u, v = _uvgrid(nu, nv)
    th = 2 * np.pi * u
    n = np.stack([np.cos(th), np.zeros_like(th), np.sin(th)], -1).astype(np.float32)
    p = r * n + np.stack([np.zeros_like(th), y0 + (y1 - y0) * v, np.zeros_like(th)], -1).astype(np.float32)
    t = np.stack([-np.sin(th), np.zeros_like(th), np.cos(th)], -1).astype(np.float32)
    b = np.broadcast_to(np.array([0, 1, 0], np.float32), p.shape).copy()
    if inward:
        n = -n
    return grid_mesh(p, n, t, b, np.stack([u * uv_scale[0], v * uv_scale[1]], -1))


def capsule(r=0.4, h=1.0, nu=64, nv=48):
    u, v = _uvgrid(nu, nv)
    th = 2 * np.pi * u
    ph = np.pi * (0.001 + 0.998 * v)
    yoff = np.where(v < 0.5, h / 2, -h / 2).astype(np.float32)
    n = np.stack([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)], -1).astype(np.float32)
    p = r * n + np.stack([np.zeros_like(th), yoff, np.zeros_like(th)], -1)
    t = np.stack([-np.sin(th), np.zeros_like(th), np.cos(th)], -1).astype(np.float32)
    b = np.stack([np.cos(ph) * np.cos(th), -np.sin(ph), np.cos(ph) * np.sin(th)], -1).astype(np.float32)
    return grid_mesh(p, n, t, b, np.stack([u * 2, v * 2], -1))


def quad(origin, eu, ev, nu, nv, uv_scale=(1.0, 1.0)):
    """planar grid origin + u*eu + v*ev; front face towards eu x ev."""
    origin, eu, ev = (np.asarray(a, np.float32) for a in (origin, eu, ev))
    u, v = _uvgrid(nu, nv)
    p = origin + u[..., None] * eu + v[..., None] * ev
    n = _unit(np.cross(eu, ev)).astype(np.float32)
    shp = p.shape
    return grid_mesh(p.astype(np.float32), np.broadcast_to(n, shp).copy(), np.broadcast_to(_unit(eu), shp).copy(),
                     np.broadcast_to(_unit(ev), shp).copy(), np.stack([u * uv_scale[0], v * uv_scale[1]], -1))


def merge(meshes):
    vs, is_, off = [], [], 0
    for v, i in meshes:
        vs.append(v)
        is_.append(i + off)
        off += len(v)
    return np.concatenate(vs), np.concatenate(is_).astype(np.uint32)


def box(sx, sy, sz, n=8, uv=2.0):
    """axis-aligned box centred at the origin, outward faces."""
    hx, hy, hz = sx / 2, sy / 2, sz / 2
    f = [quad((-hx, -hy, hz), (sx, 0, 0), (0, sy, 0), n, n, (uv, uv)),      # +z
         quad((hx, -hy, -hz), (-sx, 0, 0), (0, sy, 0), n, n, (uv, uv)),     # -z
         quad((hx, -hy, hz), (0, 0, -sz), (0, sy, 0), n, n, (uv, uv)),      # +x
         quad((-hx, -hy, -hz), (0, 0, sz), (0, sy, 0), n, n, (uv, uv)),     # -x
         quad((-hx, hy, hz), (sx, 0, 0), (0, 0, -sz), n, n, (uv, uv)),      # +y
         quad((-hx, -hy, -hz), (sx, 0, 0), (0, 0, sz), n, n, (uv, uv))]     # -y
    return merge(f)


# ----------------------------------------------------------------------------
# scene containers
# ----------------------------------------------------------------------------
class SyntheticScene:
    """everything App::load_scene would have produced, plus the create-time sizes."""

    def __init__(self, name, width, height, shadow_size, max_lights, materials, meshes, desc, lights, settings):
        self.name, self.width, self.height, self.shadow_size, self.max_lights = name, width, height, shadow_size, max_lights
        self.materials = materials   # list of (diffuse, normal, mr)
        self.meshes = meshes         # list of (vertices, indices, material_idx)
        self.desc, self.lights, self.settings = desc, lights, settings
        self.environment = None      # optional (h, w, 4) float32 equirect map for the skybox (stbi_loadf stand-in)

    @property
    def n_triangles(self):
        per_mesh = [len(i) // 3 for _, i, _ in self.meshes]
        return int(sum(per_mesh[int(o["mesh_idx"])] for o in self.desc.objects))

    def upload(self, renderer):
        """drive any object with the Renderer surface (HIP binding or oracle) like load_scene does."""
        for d, n, m in self.materials:
            renderer.create_material(d, n, m)
        for v, i, mat in self.meshes:
            renderer.create_mesh(v, i, mat)
        renderer.update_lights(self.lights)
        if self.environment is not None:
            renderer.create_hdri(self.environment)
        return renderer


def synthetic_hdri(w=512, h=256, seed=SEED + 9, sun_peak=60.0):
    """equirect RGBA32F environment (what stbi_loadf hands create_hdri, renderer.cpp:113-119): sky gradient over a dark
    ground, value-noise clouds and a small very bright sun disc, so the bilinear filter sees real HDR contrast."""
    rng = np.random.default_rng(seed)
    v = (np.arange(h, dtype=np.float32) + 0.5) / h
    u = (np.arange(w, dtype=np.float32) + 0.5) / w
    elev = (0.5 - v)[:, None] * np.float32(np.pi)                    # +pi/2 at the top row
    az = (u[None, :] - 0.5) * np.float32(2 * np.pi)
    t = np.clip(np.sin(elev), -1, 1)
    sky = np.stack([0.25 + 0.35 * (1 - t), 0.45 + 0.3 * (1 - t), 0.9 + 0.0 * t], -1)
    ground = np.stack([0.12 + 0 * t, 0.10 + 0 * t, 0.08 + 0 * t], -1)
    img = np.where((t > 0)[..., None], sky, ground).astype(np.float32) * np.ones((h, w, 1), np.float32)
    g = rng.random((h // 16 + 2, w // 16 + 2)).astype(np.float32)
    yy, xx = np.meshgrid(np.arange(h) / 16.0, np.arange(w) / 16.0, indexing="ij")
    y0, x0 = yy.astype(int), xx.astype(int)
    fy, fx = (yy - y0).astype(np.float32), (xx - x0).astype(np.float32)
    noise = (g[y0, x0] * (1 - fx) + g[y0, x0 + 1] * fx) * (1 - fy) + (g[y0 + 1, x0] * (1 - fx) + g[y0 + 1, x0 + 1] * fx) * fy
    img *= (0.7 + 0.6 * noise)[..., None]
    sun_el, sun_az = np.float32(np.deg2rad(35.0)), np.float32(np.deg2rad(40.0))
    cosang = np.sin(elev) * np.sin(sun_el) + np.cos(elev) * np.cos(sun_el) * np.cos(az - sun_az)
    img += (np.float32(sun_peak) * np.exp((np.clip(cosang, -1, 1) - 1) * 900.0))[..., None] * np.array([1.0, 0.9, 0.7], np.float32)
    out = np.ones((h, w, 4), np.float32)
    out[..., :3] = img
    return np.ascontiguousarray(out)


DEFAULT_SUN = dict(position=(-10.0, 32.0, -2.48), rotation=(-70.0, 12.0), color=(8.0, 8.0, 8.0))   # src/app.hpp:51-55


def random_lights(rng, n, lo, hi, intensity=10.0):
    """n lights uniform in the box [lo,hi]; colour = random hue x intensity (cf. src/app.cpp:515-518)."""
    pos = (np.asarray(lo, np.float32) + rng.random((n, 3), dtype=np.float32) * (np.asarray(hi, np.float32) - np.asarray(lo, np.float32)))
    col = np.array([colorsys.hsv_to_rgb(h, 1.0, 1.0) for h in rng.random(n)], np.float32).reshape(n, 3) * intensity
    return make_lights(pos, col)


def config1(scale=1.0, tex=None):
    """SciFiHelmet stand-in: 512x512, UV-sphere r=1 (128x64), one material, camera/sun read off the
    reference's scifi-helmet.png overlay, 1 directional light, no shadow map, Reinhard."""
    rng = np.random.default_rng(SEED + 1)
    size = max(16, int(round(512 * scale)) // 8 * 8)
    tex = tex or max(64, int(2048 * scale))
    mats = [make_material_textures(rng, tex)]
    seg = max(16, int(128 * scale))
    meshes = [uv_sphere(1.0, seg, seg // 2, (4.0, 2.0)) + (0,)]
    desc = SceneDesc(camera=dict(eye=(-3.043, 1.322, 4.675), rotation=(-11.0, -60.5), aspect=1.0, fov_y=45.0, z_near_far=(0.1, 1000.0)),
                     ambient=0.1, sun=dict(position=(-10.0, 32.0, -2.48), rotation=(-46.9, -32.3), color=(8.0, 8.0, 8.0)),
                     objects=make_objects([(np.eye(4), 0)]))
    return SyntheticScene("config1-scifihelmet-standin", size, size, 0, 16, mats, meshes, desc,
                          np.zeros(0, LIGHT_DTYPE), (TM_REINHARD, 2.2, 1.0))


def config2(scale=1.0, tex=None):
    """FlightHelmet stand-in: 1080p, 6 meshes / 6 materials on a ground quad, default sun, 2048^2 shadow map, Reinhard."""
    rng = np.random.default_rng(SEED + 2)
    w, h = max(16, int(round(1920 * scale)) // 8 * 8), max(8, int(round(1080 * scale)) // 8 * 8)
    tex = tex or max(64, int(2048 * scale))
    S = max(64, int(2048 * scale))
    q = max(0.15, scale)
    mats = [make_material_textures(rng, tex) for _ in range(6)]
    meshes = [uv_sphere(0.6, int(128 * q), int(64 * q), (2.0, 1.0)) + (0,),
              torus(0.9, 0.25, int(96 * q), int(48 * q)) + (1,),
              capsule(0.35, 0.9, int(64 * q), int(48 * q)) + (2,),
              box(1.6, 0.4, 1.6, max(2, int(8 * q))) + (3,),
              cylinder(0.3, 0.0, 1.6, int(48 * q), int(24 * q), (2.0, 2.0)) + (4,),
              quad((-4, 0, 4), (8, 0, 0), (0, 0, -8), max(2, int(32 * q)), max(2, int(32 * q)), (4.0, 4.0)) + (5,)]
    objs = make_objects([(translation(0.0, 1.6, 0.0), 0), (translation(0.0, 0.75, 0.0), 1),
                         (translation(1.8, 0.8, 0.6) @ rotation_y(30), 2), (translation(0.0, 0.2, 0.0), 3),
                         (translation(-1.7, 0.0, -0.8), 4), (np.eye(4), 5)])
    desc = SceneDesc(camera=dict(eye=(-3.2, 2.2, 4.2), rotation=(-18.0, -52.0), aspect=w / h, fov_y=45.0, z_near_far=(0.1, 1000.0)),
                     ambient=0.1, sun=DEFAULT_SUN, objects=objs)
    return SyntheticScene("config2-flighthelmet-standin", w, h, S, 16, mats, meshes, desc,
                          np.zeros(0, LIGHT_DTYPE), (TM_REINHARD, 2.2, 1.0))


def atrium(width, height, shadow_size, n_lights, scale=1.0, tex=None, tm=TM_ACES, seed=3, name="atrium"):
    """Sponza stand-in (configs 3-5): closed 30 x 14 x 12 m hall, two storeys of 2 x 10 columns, balconies,
    ceiling with a 20 x 6 m opening the sun shines through; camera inside -> 100 % pixel coverage,
    near-plane and guard-band clipping exercised, overdraw 2-3x.  UV tiling repeats every 4 m, i.e. 1024^2 textures are
    sampled at ~1 texel per pixel at 4K from ~10 m ("~1:1 texel density"; the reference has one mip level,
    rhi.cpp:550, so anything denser is pure minification overfetch)."""
    rng = np.random.default_rng(SEED + seed)
    tex = tex or max(64, int(1024 * scale))
    q = max(0.1, scale)
    g = lambda metres: max(1, int(round(metres * 4 * q)))     # 0.25 m grid at scale 1
    mats = [make_material_textures(rng, tex) for _ in range(25)]
    meshes, objs = [], []

    def add(mesh, mat, trs=None):
        meshes.append(mesh + (mat,))
        objs.append((np.eye(4, dtype=np.float32) if trs is None else trs, len(meshes) - 1))

    X, Z, H = 15.0, 7.0, 12.0
    add(quad((-X, 0, Z), (2 * X, 0, 0), (0, 0, -2 * Z), g(30), g(14), (7.5, 3.5)), 0)                  # floor (+y)
    # ceiling (-y) around the opening x in [-10,10], z in [-3,3]
    ceil = [quad((-X, H, -Z), (2 * X, 0, 0), (0, 0, 4.0), g(30), g(4), (7.5, 1.0)),
            quad((-X, H, 3.0), (2 * X, 0, 0), (0, 0, 4.0), g(30), g(4), (7.5, 1.0)),
            quad((-X, H, -3.0), (5.0, 0, 0), (0, 0, 6.0), g(5), g(6), (1.25, 1.5)),
            quad((10.0, H, -3.0), (5.0, 0, 0), (0, 0, 6.0), g(5), g(6), (1.25, 1.5))]
    add(merge(ceil), 1)
    add(quad((X, 0, -Z), (0, 0, 2 * Z), (0, H, 0), g(14), g(12), (3.5, 3.0)), 2)      # far wall x=+15 (faces -x)
    add(quad((-X, 0, Z), (0, 0, -2 * Z), (0, H, 0), g(14), g(12), (3.5, 3.0)), 3)     # back wall x=-15 (faces +x)
    add(quad((X, 0, Z), (-2 * X, 0, 0), (0, H, 0), g(30), g(12), (7.5, 3.0)), 4)     # wall z=+7 (faces -z)
    add(quad((-X, 0, -Z), (2 * X, 0, 0), (0, H, 0), g(30), g(12), (7.5, 3.0)), 5)    # wall z=-7 (faces +z)
    for s, mat in ((1.0, 6), (-1.0, 7)):                                              # balconies at y = 6
        z_in, z_out = 4.0 * s, Z * s
        top = quad((-X, 6.0, max(z_in, z_out)), (2 * X, 0, 0), (0, 0, -3.0), g(30), g(3), (7.5, 0.75))
        bot = quad((-X, 5.7, min(z_in, z_out)), (2 * X, 0, 0), (0, 0, 3.0), g(30), g(3), (7.5, 0.75))
        edge = (quad((X, 5.7, z_in), (-2 * X, 0, 0), (0, 0.3, 0), g(30), 1, (7.5, 0.075)) if s > 0 else
                quad((-X, 5.7, z_in), (2 * X, 0, 0), (0, 0.3, 0), g(30), 1, (7.5, 0.075)))
        add(merge([top, bot, edge]), mat)
    col_lo = cylinder(0.35, 0.0, 5.7, max(8, int(32 * q)), max(2, int(64 * q)), (0.5, 1.5))
    col_hi = cylinder(0.30, 6.0, 12.0, max(8, int(32 * q)), max(2, int(64 * q)), (0.5, 1.5))
    k = 0
    for base in (col_lo, col_hi):                                                     # one mesh per column: a mesh owns its material
        for zrow in (-4.5, 4.5):
            for i in range(10):
                add(base, 8 + (k % 17), translation(-13.5 + 3.0 * i, 0.0, zrow))
                k += 1
    lights = random_lights(rng, n_lights, (-X + 0.5, 0.5, -Z + 0.5), (X - 0.5, H - 0.5, Z - 0.5))
    desc = SceneDesc(camera=dict(eye=(0.0, 5.0, 0.0), rotation=(-15.0, 0.0), aspect=width / height, fov_y=45.0,
                                 z_near_far=(0.1, 1000.0)),
                     ambient=0.1, sun=DEFAULT_SUN, objects=make_objects(objs), point_lights=lights)
    return SyntheticScene(name, width, height, shadow_size, max(16, n_lights), mats, meshes, desc, lights, (tm, 2.2, 1.0))


def _dims(w, h, scale):
    return max(16, int(round(w * scale)) // 8 * 8), max(8, int(round(h * scale)) // 8 * 8)


def config3(scale=1.0, tex=None):
    """Sponza stand-in, 4K, 1 dir + 64 point lights, 4000^2 shadow map, ACES: the config BASELINE.json's metric is quoted on."""
    w, h = _dims(3840, 2160, scale)
    return atrium(w, h, max(64, int(4000 * scale)), 64, scale, tex, name="config3-sponza-standin-64")


def config4(scale=1.0, tex=None):
    w, h = _dims(3840, 2160, scale)
    return atrium(w, h, max(64, int(4000 * scale)), 256, scale, tex, name="config4-sponza-standin-256")


def config5(scale=1.0, tex=None):
    w, h = _dims(7680, 4320, scale)
    sc = atrium(w, h, max(64, int(4096 * scale)), 1024, scale, tex, name="config5-sponza-standin-1024")
    sc.environment = synthetic_hdri(max(64, int(2048 * scale)) // 2 * 2, max(32, int(1024 * scale)) // 2 * 2)   # "HDR env" of BASELINE configs[4]
    return sc


CONFIGS = {1: config1, 2: config2, 3: config3, 4: config4, 5: config5}


def random_gbuffer(rng, rows, width, n_materials, coverage=1.0, world_lo=(-15, 0, -7), world_hi=(15, 12, 7),
                   light_proj_view=None):
    """random (not constant) G-buffer attributes for full-size shading runs without a rasteriser:
    uv in [0,4), orthonormal-ish TBN, world position in a box, light-space position = light_proj_view * world
    (or uniform in the light frustum).  Returns (attrs[rows,width,18] f32, material[rows,width] u32)."""
    a = np.empty((rows, width, 18), np.float32)
    a[..., 0:2] = rng.random((rows, width, 2), dtype=np.float32) * 4.0
    n = _unit(rng.standard_normal((rows, width, 3)).astype(np.float32))
    t = _unit(np.cross(n, _unit(rng.standard_normal((rows, width, 3)).astype(np.float32))))
    a[..., 2:5], a[..., 5:8], a[..., 8:11] = t, np.cross(n, t), n
    lo, hi = np.asarray(world_lo, np.float32), np.asarray(world_hi, np.float32)
    world = lo + rng.random((rows, width, 3), dtype=np.float32) * (hi - lo)
    a[..., 11:14] = world
    if light_proj_view is not None:
        m = np.asarray(light_proj_view, np.float32).reshape(4, 4)   # [col][row]
        w4 = np.concatenate([world, np.ones((rows, width, 1), np.float32)], -1)
        a[..., 14:18] = np.einsum("cr,hwc->hwr", m, w4)
    else:
        a[..., 14:16] = rng.random((rows, width, 2), dtype=np.float32) * 2.2 - 1.1
        a[..., 16] = rng.random((rows, width), dtype=np.float32) * 1.1
        a[..., 17] = 1.0
    mat = rng.integers(0, n_materials, (rows, width), dtype=np.uint32)
    if coverage < 1.0:
        mat[rng.random((rows, width)) >= coverage] = 0xFFFFFFFF
    return a, mat
