"""Deterministic synthetic stand-ins for the reference's git-LFS assets.

Every mesh / texture under /root/reference/renderer/assets is a 130-byte LFS pointer stub
(SURVEY.md F2), so the Cornell room, the "bunny" and "dragon" heroes and the bunny-material-0
textures are re-authored here as seeded procedural generators.  Topology and naming follow the
reference scene files (renderer/src/scene/scene_3.rs:13-115: bunny, box, hidari/left, migi/right,
yuka/floor, oku/back, tenjou/ceiling, light), sizes follow SURVEY.md §8(d).

`load_obj_semantics` mirrors what TriangleMesh::load_obj derives from an OBJ file
(scene/src/geometry/impls/triangle_mesh.rs:141-242): unit normals and, when texcoords exist,
one tangent per triangle with the reference's fallback rules.

Room (world space, +Y up, camera of scene_3.rs:110-114 at (0, 3.15221, 6) looking (0,-0.9,-3.2)):
  x in [-4, 4], y in [0, 4], z in [-4, 2]; the front (z = 2) is open, as in the reference's asset
  list (no front wall).  At vfov 45 deg the 16:9 frustum enters through the opening (half extents
  2.95 x 1.66 at z = 2 < 4 x 4), so every camera ray lands inside the room; side walls, floor,
  back wall and the rear part of the ceiling (with the light) are all in view.
"""
import numpy as np

F = np.float32


def _quad(p0, p1, p2, p3, n):
    """Two triangles p0-p1-p2, p0-p2-p3 with a constant normal, no UVs (like the room OBJs)."""
    pos = np.array([p0, p1, p2, p3], dtype=F)
    nrm = np.tile(np.array(n, dtype=F), (4, 1))
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    return dict(pos=pos, nrm=nrm, uv=None, idx=idx)


def _cuboid(lo, hi, rot_y_deg=0.0, center=None):
    """12-triangle box with per-face normals (24 vertices), optional rotation about +Y."""
    lo = np.array(lo, dtype=np.float64); hi = np.array(hi, dtype=np.float64)
    faces = [  # (axis, sign)
        (0, -1), (0, 1), (1, -1), (1, 1), (2, -1), (2, 1)]
    pos, nrm, idx = [], [], []
    for axis, sgn in faces:
        u, v = [(1, 2), (2, 0), (0, 1)][axis]
        c = hi[axis] if sgn > 0 else lo[axis]
        corners = []
        for (a, b) in [(0, 0), (1, 0), (1, 1), (0, 1)]:
            p = np.zeros(3); p[axis] = c
            p[u] = hi[u] if a else lo[u]
            p[v] = hi[v] if b else lo[v]
            corners.append(p)
        if sgn < 0:
            corners = corners[::-1]
        n = np.zeros(3); n[axis] = sgn
        base = len(pos)
        pos += corners; nrm += [n] * 4
        idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    pos = np.array(pos); nrm = np.array(nrm)
    if rot_y_deg:
        a = np.deg2rad(rot_y_deg); ca, sa = np.cos(a), np.sin(a)
        R = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]])
        c = np.array(center if center is not None else (lo + hi) / 2)
        pos = (pos - c) @ R.T + c
        nrm = nrm @ R.T
    return dict(pos=pos.astype(F), nrm=nrm.astype(F), uv=None, idx=np.array(idx, dtype=np.uint32))


def cornell_room():
    """The seven room meshes of scene_3.rs:33-106 keyed by the reference's OBJ names."""
    x0, x1, y0, y1, z0, z1 = -4.0, 4.0, 0.0, 4.0, -4.0, 2.0
    return {
        "yuka": _quad((x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0), (0, 1, 0)),      # floor
        "tenjou": _quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), (0, -1, 0)),   # ceiling
        "oku": _quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (0, 0, 1)),       # back
        "hidari": _quad((x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1), (1, 0, 0)),    # left (red)
        "migi": _quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), (-1, 0, 0)),     # right (green)
        "box": _cuboid((1.0, 0.0, -2.2), (2.4, 2.0, -0.8), rot_y_deg=-20.0),
        # light.obj: thin two-sided emissive slab just under the ceiling (12 triangles)
        "light": _cuboid((-1.0, 3.90, -2.5), (1.0, 3.98, -1.0)),
    }


def _smooth_noise_sphere(theta, phi, rng, n_lobes, amp, sharp):
    """Sum of Gaussian-like bumps on the unit sphere (keeps the surface star-shaped)."""
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], -1)
    r = np.zeros(theta.shape)
    centers = rng.normal(size=(n_lobes, 3)); centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    amps = rng.uniform(-0.5, 1.0, size=n_lobes) * amp
    sharps = rng.uniform(0.5, 1.5, size=n_lobes) * sharp
    for c, a, s in zip(centers, amps, sharps):
        r += a * np.exp(s * (d @ c - 1.0))
    return r


def blob_mesh(n_lon, n_bands, seed, lobes, center, scale, with_uv=True):
    """Closed genus-0 star-shaped mesh: lat/long grid with a duplicated seam column (spherical UVs),
    single-vertex poles, n_lon*(2*n_bands-2) triangles, area-weighted smooth normals."""
    rng = np.random.default_rng(seed)
    # interior rings 1..n_bands-1, poles handled separately
    th = np.linspace(0.0, np.pi, n_bands + 1)[1:-1]
    ph = np.linspace(0.0, 2.0 * np.pi, n_lon + 1)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    big = _smooth_noise_sphere(TH, PH % (2 * np.pi), np.random.default_rng(seed), lobes[0], lobes[1], lobes[2])
    fine = _smooth_noise_sphere(TH, PH % (2 * np.pi), np.random.default_rng(seed + 1), lobes[3], lobes[4], lobes[5])
    R = 1.0 + big + fine
    R[:, -1] = R[:, 0]   # exact seam closure
    d = np.stack([np.sin(TH) * np.cos(PH), np.cos(TH), np.sin(TH) * np.sin(PH)], -1)
    d[:, -1, :] = d[:, 0, :]
    P = d * R[..., None] * np.array([1.0, 1.15, 0.85])
    pole_r = [1.0 + float(np.mean(R[0])) * 0.0 + float(np.mean(big[0] + fine[0])), 1.0 + float(np.mean(big[-1] + fine[-1]))]
    top = np.array([0.0, pole_r[0] * 1.15, 0.0]); bot = np.array([0.0, -pole_r[1] * 1.15, 0.0])
    nr, nc = P.shape[0], P.shape[1]          # nr = n_bands-1 rings, nc = n_lon+1 columns
    pos = np.concatenate([P.reshape(-1, 3), np.tile(top, (n_lon, 1)), np.tile(bot, (n_lon, 1))], 0)
    U = np.tile(np.linspace(0.0, 1.0, nc), (nr, 1))
    V = 1.0 - np.tile((th / np.pi)[:, None], (1, nc))
    uv = np.concatenate([np.stack([U, V], -1).reshape(-1, 2),
                         np.stack([(np.arange(n_lon) + 0.5) / n_lon, np.ones(n_lon)], -1),
                         np.stack([(np.arange(n_lon) + 0.5) / n_lon, np.zeros(n_lon)], -1)], 0)
    vid = lambda r, c: r * nc + c
    top0 = nr * nc; bot0 = top0 + n_lon
    tris = []
    for c in range(n_lon):
        tris.append((top0 + c, vid(0, c + 1), vid(0, c)))
        tris.append((bot0 + c, vid(nr - 1, c), vid(nr - 1, c + 1)))
    for r in range(nr - 1):
        for c in range(n_lon):
            a, b, e, f = vid(r, c), vid(r, c + 1), vid(r + 1, c), vid(r + 1, c + 1)
            tris.append((a, b, f)); tris.append((a, f, e))
    idx = np.array(tris, dtype=np.uint32)
    # orient outward
    p0, p1, p2 = pos[idx[:, 0]], pos[idx[:, 1]], pos[idx[:, 2]]
    fn = np.cross(p1 - p0, p2 - p0)
    cen = (p0 + p1 + p2) / 3.0
    flip = np.sum(fn * cen, 1) < 0
    idx[flip] = idx[flip][:, [0, 2, 1]]
    fn[flip] *= -1
    # smooth normals accumulated per unique position (seam + poles share)
    key = np.round(pos, 9)
    _, inv = np.unique(key, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    acc = np.zeros((inv.max() + 1, 3))
    for k in range(3):
        np.add.at(acc, inv[idx[:, k]], fn)
    nrm = acc[inv]
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    pos = pos * scale + np.array(center)
    m = dict(pos=pos.astype(F), nrm=nrm.astype(F), uv=uv.astype(F) if with_uv else None, idx=idx.reshape(-1).astype(np.uint32))
    return m


def bunny_class():
    """7 168-triangle hero of scenes 3/8/10 (stands in for bunny.obj, 530 KB)."""
    return blob_mesh(64, 57, seed=11, lobes=(10, 0.35, 6.0, 24, 0.06, 30.0), center=(-1.2, 1.25, -1.6), scale=1.0)


def dragon_class():
    """20 480-triangle hero of scene 17 (stands in for dragon.min.obj, 1.5 MB), modelled around the origin:
    the instance transform of scene_17.rs:61-69 places it."""
    return blob_mesh(128, 81, seed=23, lobes=(14, 0.30, 8.0, 60, 0.05, 60.0), center=(0.0, 0.5, 0.0), scale=0.42)


def single_triangle():
    """The SingleTrianglePrimitive of scene_1.rs:46-67 as a one-triangle mesh (positions, +z normals, uvs)."""
    pos = np.array([[-2.0, 0.0, 0.0], [2.0, 0.0, 0.0], [-2.0, 4.0, 0.0]], dtype=F)
    nrm = np.array([[0, 0, 1]] * 3, dtype=F)
    uv = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]], dtype=F)
    return dict(pos=pos, nrm=nrm, uv=uv, idx=np.array([[0, 1, 2]], dtype=np.uint32))


def sky_envmap(w=128, h=64, seed=3):
    """Synthetic lat-long HDR sky standing in for scene_19.rs's scythian_tombs_2_1k.exr (an LFS stub): blue-to-white sky
    gradient, warm sun disc (~60x the sky), dim ground, mild seeded clouds.  Row 0 = +y pole, float32 (h, w, 3)."""
    rng = np.random.default_rng(seed)
    v = (np.arange(h, dtype=np.float64) + 0.5) / h
    u = (np.arange(w, dtype=np.float64) + 0.5) / w
    theta, phi = np.meshgrid(v * np.pi, u * 2 * np.pi, indexing="ij")
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], -1)
    up = np.clip(d[..., 1], 0.0, 1.0)
    sky = (1.0 - up[..., None]) * np.array([0.9, 0.95, 1.0]) + up[..., None] * np.array([0.25, 0.45, 0.9])
    ground = np.array([0.12, 0.10, 0.08])
    img = np.where(d[..., 1:2] >= 0.0, sky, ground)
    sun_dir = np.array([0.5, 0.6, 0.62]); sun_dir /= np.linalg.norm(sun_dir)
    c = np.clip(d @ sun_dir, -1.0, 1.0)
    img = img + np.array([60.0, 52.0, 40.0]) * np.exp((c - 1.0) * 250.0)[..., None]
    clouds = _value_noise(max(w, h), 4, rng)[:h, :w]
    img = img * (0.85 + 0.3 * clouds[..., None])
    return np.ascontiguousarray(img.astype(F))


def _value_noise(n, octaves, rng):
    out = np.zeros((n, n))
    amp, tot = 1.0, 0.0
    for o in range(octaves):
        cells = 4 * 2 ** o
        g = rng.random((cells, cells))
        xs = np.arange(n) / n * cells
        i0 = np.floor(xs).astype(int) % cells; i1 = (i0 + 1) % cells
        t = xs - np.floor(xs); t = t * t * (3 - 2 * t)
        a = g[i0][:, i0] * (1 - t)[None, :] + g[i0][:, i1] * t[None, :]
        b = g[i1][:, i0] * (1 - t)[None, :] + g[i1][:, i1] * t[None, :]
        out += amp * (a * (1 - t)[:, None] + b * t[:, None])
        tot += amp; amp *= 0.5
    return out / tot


def bunny_textures(size=1024, seed=5):
    """BaseColor.png / Normal.png stand-ins: tileable value-noise albedo (sRGB-encoded RGB8) and a
    bump-derived tangent-space normal map (OpenGL +Y, used with flip_y=false, scene_3.rs:22-25)."""
    rng = np.random.default_rng(seed)
    h = _value_noise(size, 5, rng)
    tint = np.stack([_value_noise(size, 3, rng), _value_noise(size, 3, rng), _value_noise(size, 3, rng)], -1)
    base = np.array([0.78, 0.62, 0.48])
    alb = np.clip(base * (0.55 + 0.45 * h[..., None]) * (0.8 + 0.4 * tint), 0.02, 0.98)
    albedo = (alb * 255.0 + 0.5).astype(np.uint8)
    bump = _value_noise(size, 6, rng)
    gy, gx = np.gradient(bump)
    k = 6.0 * size / 64.0
    n = np.stack([-gx * k, gy * k, np.ones_like(gx)], -1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    normal = ((n * 0.5 + 0.5) * 255.0 + 0.5).astype(np.uint8)
    return np.ascontiguousarray(albedo), np.ascontiguousarray(normal)


def material_maps(size=1024, seed=9):
    """Grey Metallic / Roughness / ClearcoatThickness stand-ins (dragon-material/*.png), replicated to RGB8."""
    rng = np.random.default_rng(seed)
    met = (_value_noise(size, 3, rng) > 0.5).astype(np.float64) * 0.9 + 0.05
    rgh = 0.15 + 0.6 * _value_noise(size, 4, rng)
    thk = 0.2 + 0.7 * _value_noise(size, 3, rng)
    grey = lambda a: np.ascontiguousarray(np.repeat((np.clip(a, 0, 1) * 255.0 + 0.5).astype(np.uint8)[..., None], 3, -1))
    return grey(met), grey(rgh), grey(thk)


def load_obj_semantics(mesh):
    """What TriangleMesh::load_obj computes on top of raw OBJ arrays
    (geometry/impls/triangle_mesh.rs:154-242): per-triangle tangents when texcoords exist."""
    pos, uv, idx = mesh["pos"], mesh["uv"], mesh["idx"].reshape(-1, 3)
    out = dict(mesh)
    out["tangent"] = None
    if uv is None:
        return out
    p0, p1, p2 = pos[idx[:, 0]], pos[idx[:, 1]], pos[idx[:, 2]]
    e1 = (p1 - p0).astype(F); e2 = (p2 - p0).astype(F)
    d1 = (uv[idx[:, 1]] - uv[idx[:, 0]]).astype(F); d2 = (uv[idx[:, 2]] - uv[idx[:, 0]]).astype(F)
    den = (d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0]).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (F(1.0) / den).astype(F)
        t = (r[:, None] * (e1 * d2[:, 1:2] - e2 * d1[:, 1:2])).astype(F)
        ln = np.sqrt((t[:, 0] * t[:, 0] + t[:, 1] * t[:, 1] + t[:, 2] * t[:, 2]).astype(F)).astype(F)
        tn = (t * (F(1.0) / ln)[:, None]).astype(F)
    # fallback_tangent (triangle_mesh.rs:203-214)
    cr = np.cross(e1, e2).astype(F)
    cl2 = np.sum(cr * cr, 1).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        nn = (cr * (F(1.0) / np.sqrt(cl2))[:, None]).astype(F)
    cand = np.where((np.abs(nn[:, 0]) > F(0.999))[:, None], np.array([0, 1, 0], F), np.array([1, 0, 0], F)).astype(F)
    proj = np.sum(nn * cand, 1, keepdims=True).astype(F)
    fb = (cand - nn * proj).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        fb = (fb * (F(1.0) / np.sqrt(np.sum(fb * fb, 1, keepdims=True)))).astype(F)
    fb = np.where((cl2 < F(1e-12))[:, None], np.array([1, 0, 0], F), fb)
    use_fb = (np.abs(den) < F(1e-6)) | np.isnan(tn).any(1)
    out["tangent"] = np.where(use_fb[:, None], fb, tn).astype(F)
    return out
