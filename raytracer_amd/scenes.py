"""Seeded procedural scenes, as arrays in the reference's in-memory Triangle layout.

The reference ships no meshes, textures or sample scenes (its config.txt is a commented
template with placeholder paths), so every workload here is generated: a small mixed-material
scene for parity tests, a bunny-class closed mesh (~69.5k triangles), and a Sponza-class
textured atrium with a cube-map sky (~262k triangles).

Triangles follow what RayTracer::loadModel produces (/root/reference/src/RayTracer.cpp:78-105):
positions scaled+offset, per-vertex normals, planeNormal = normalize((v1-v0) x (v2-v0)),
one colour / material / texture per part.
"""
import numpy as np

TRI_DTYPE = np.dtype([
    ("v", "<f4", (3, 4)), ("n", "<f4", (3, 4)), ("pn", "<f4", (4,)), ("color", "<f4", (4,)),
    ("uv", "<f4", (3, 2)), ("rough", "<f4"), ("ior", "<f4"), ("tex", "<i4"),
    ("emit", "u1"), ("metal", "u1"), ("transp", "u1"), ("pad", "u1", (9,)),
])
NODE_DTYPE = np.dtype([
    ("bmin", "<f4", (4,)), ("bmax", "<f4", (4,)), ("left", "<i4"), ("right", "<i4"), ("tri", "<i4"), ("pad", "<i4"),
])

F = np.float32


def _normalize_rows(v):
    """Row-wise v / sqrt(|v|^2) in float32 with Eigen's reduction order (x^2+z^2)+(y^2+0)."""
    v = v.astype(F)
    z = (v[:, 0] * v[:, 0] + v[:, 2] * v[:, 2]) + (v[:, 1] * v[:, 1] + F(0))
    out = v.copy()
    ok = z > 0
    out[ok] = v[ok] / np.sqrt(z[ok])[:, None]
    return out


def _cross(a, b):
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1],
                     a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                     a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1).astype(F)


class Material:
    def __init__(self, color=(0.8, 0.8, 0.8), metal=False, emit=False, transp=False, rough=0.0, ior=1.0, tex=-1):
        self.color, self.metal, self.emit, self.transp = color, metal, emit, transp
        self.rough, self.ior, self.tex = rough, ior, tex


class SceneBuilder:
    """Collects indexed mesh parts and emits TRI_DTYPE records."""

    def __init__(self):
        self.parts = []
        self.textures = []   # list of (H,W,3) uint8, as held in memory by the reference (row 0 = v 0)
        self.sky = None      # (brightness, [6 texture ids])

    def add_texture(self, rgb):
        self.textures.append(np.ascontiguousarray(rgb, np.uint8))
        return len(self.textures) - 1

    def set_skybox(self, brightness, faces):
        ids = [self.add_texture(f) for f in faces]
        self.sky = (float(brightness), ids)

    def add_mesh(self, verts, faces, mat, uv=None, smooth=True, flip=False):
        verts = np.asarray(verts, F)
        faces = np.asarray(faces, np.int64)
        if flip:
            faces = faces[:, ::-1]
        v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
        fn_raw = _cross((v1 - v0).astype(F), (v2 - v0).astype(F))
        pn = _normalize_rows(fn_raw)
        keep = np.isfinite(pn).all(axis=1) & ((pn * pn).sum(axis=1) > 0.5)
        faces, pn, fn_raw = faces[keep], pn[keep], fn_raw[keep]
        if smooth:
            vn = np.zeros_like(verts, dtype=np.float64)
            for k in range(3):
                np.add.at(vn, faces[:, k], fn_raw.astype(np.float64))
            vn = _normalize_rows(vn.astype(F))
        n = len(faces)
        t = np.zeros(n, TRI_DTYPE)
        for k in range(3):
            t["v"][:, k, :3] = verts[faces[:, k]]
            t["n"][:, k, :3] = vn[faces[:, k]] if smooth else pn
            if uv is not None:
                t["uv"][:, k, :] = np.asarray(uv, F)[faces[:, k]]
        t["pn"][:, :3] = pn
        t["color"][:, :3] = np.asarray(mat.color, F)
        t["rough"] = F(mat.rough)
        t["ior"] = F(mat.ior)
        t["tex"] = mat.tex if uv is not None else -1
        t["emit"], t["metal"], t["transp"] = int(mat.emit), int(mat.metal), int(mat.transp)
        self.parts.append(t)
        return n

    # ---- primitives -------------------------------------------------------------
    def grid(self, origin, du, dv, nu, nv, mat, disp=None, smooth=True, flip=False):
        """(nu x nv) quads spanning origin + s*du + t*dv, s,t in [0,1]; disp(s,t)->offset along du x dv."""
        origin, du, dv = (np.asarray(a, F) for a in (origin, du, dv))
        s, t = np.meshgrid(np.linspace(0, 1, nu + 1, dtype=F), np.linspace(0, 1, nv + 1, dtype=F), indexing="ij")
        p = origin[None, None, :] + s[..., None] * du + t[..., None] * dv
        if disp is not None:
            nrm = np.cross(du, dv)
            nrm = (nrm / np.linalg.norm(nrm)).astype(F)
            p = p + disp(s, t)[..., None].astype(F) * nrm
        verts = p.reshape(-1, 3).astype(F)
        # uv stays inside [0,1]: the reference neither wraps nor clamps (src/Texture.cpp:45-47)
        uv = np.stack([s.reshape(-1), t.reshape(-1)], axis=1).astype(F)
        i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
        a = (i * (nv + 1) + j).reshape(-1)
        b = a + (nv + 1)
        c = b + 1
        d = a + 1
        faces = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 0)
        return self.add_mesh(verts, faces, mat, uv=uv, smooth=smooth, flip=flip)

    def box(self, lo, hi, mat, n=1, inward=False):
        lo, hi = np.asarray(lo, F), np.asarray(hi, F)
        e = hi - lo
        ex, ey, ez = np.array([e[0], 0, 0], F), np.array([0, e[1], 0], F), np.array([0, 0, e[2]], F)
        cnt = 0
        # each face wound so that du x dv points outward
        cnt += self.grid(lo, ey, ex, n, n, mat, smooth=False, flip=inward)                  # z = lo, normal -z
        cnt += self.grid(lo + ez, ex, ey, n, n, mat, smooth=False, flip=inward)             # z = hi, normal +z
        cnt += self.grid(lo, ex, ez, n, n, mat, smooth=False, flip=inward)                  # y = lo, normal -y
        cnt += self.grid(lo + ey, ez, ex, n, n, mat, smooth=False, flip=inward)             # y = hi, normal +y
        cnt += self.grid(lo, ez, ey, n, n, mat, smooth=False, flip=inward)                  # x = lo, normal -x
        cnt += self.grid(lo + ex, ey, ez, n, n, mat, smooth=False, flip=inward)             # x = hi, normal +x
        return cnt

    def sphere(self, center, radius, rings, segs, mat, disp=None, flip=False):
        """Lat-long sphere; disp(theta, phi) -> relative radius change."""
        th, ph = np.meshgrid(np.linspace(0, np.pi, rings + 1), np.linspace(0, 2 * np.pi, segs + 1), indexing="ij")
        r = np.full_like(th, radius)
        if disp is not None:
            r = r * (1.0 + disp(th, ph))
        x = r * np.sin(th) * np.cos(ph)
        y = r * np.cos(th)
        z = r * np.sin(th) * np.sin(ph)
        # weld the seam and the poles so that the mesh is closed
        x[:, -1], y[:, -1], z[:, -1] = x[:, 0], y[:, 0], z[:, 0]
        verts = (np.stack([x, y, z], -1).reshape(-1, 3) + np.asarray(center)).astype(F)
        uv = np.stack([ph / (2 * np.pi), 1.0 - th / np.pi], -1).reshape(-1, 2).astype(F)
        i, j = np.meshgrid(np.arange(rings), np.arange(segs), indexing="ij")
        a = (i * (segs + 1) + j).reshape(-1)
        b = a + (segs + 1)
        c = b + 1
        d = a + 1
        faces = np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)], 0)
        return self.add_mesh(verts, faces, mat, uv=uv, smooth=True, flip=flip)

    def cylinder(self, base, radius, height, segs, stacks, mat, flute=0.0, flutes=12):
        ph, h = np.meshgrid(np.linspace(0, 2 * np.pi, segs + 1), np.linspace(0, 1, stacks + 1), indexing="ij")
        r = radius * (1.0 + flute * np.cos(flutes * ph))
        x, z, y = r * np.cos(ph), r * np.sin(ph), h * height
        x[-1], z[-1] = x[0], z[0]
        verts = (np.stack([x, y, z], -1).reshape(-1, 3) + np.asarray(base)).astype(F)
        uv = np.stack([ph / (2 * np.pi), h], -1).reshape(-1, 2).astype(F)
        i, j = np.meshgrid(np.arange(segs), np.arange(stacks), indexing="ij")
        a = (i * (stacks + 1) + j).reshape(-1)
        b = a + (stacks + 1)
        c = b + 1
        d = a + 1
        faces = np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)], 0)
        return self.add_mesh(verts, faces, mat, uv=uv, smooth=True)

    def arch(self, c0, span_axis, radius, tube, segs, tsegs, mat):
        """Half torus standing on the floor: centre c0, opening along span_axis (0 = x, 2 = z)."""
        a, b = np.meshgrid(np.linspace(0, np.pi, segs + 1), np.linspace(0, 2 * np.pi, tsegs + 1), indexing="ij")
        rr = radius + tube * np.cos(b)
        u = rr * np.cos(a)
        y = rr * np.sin(a)
        w = tube * np.sin(b)
        w[:, -1], u[:, -1], y[:, -1] = w[:, 0], u[:, 0], y[:, 0]
        if span_axis == 0:
            p = np.stack([u, y, w], -1)
        else:
            p = np.stack([w, y, u], -1)
        verts = (p.reshape(-1, 3) + np.asarray(c0)).astype(F)
        uv = np.stack([a / np.pi, b / (2 * np.pi)], -1).reshape(-1, 2).astype(F)
        i, j = np.meshgrid(np.arange(segs), np.arange(tsegs), indexing="ij")
        q = (i * (tsegs + 1) + j).reshape(-1)
        r2 = q + (tsegs + 1)
        s2 = r2 + 1
        t2 = q + 1
        faces = np.concatenate([np.stack([q, r2, s2], 1), np.stack([q, s2, t2], 1)], 0)
        return self.add_mesh(verts, faces, mat, uv=uv, smooth=True, flip=(span_axis == 0))

    def triangles(self):
        if not self.parts:
            return np.zeros(0, TRI_DTYPE)
        return np.concatenate(self.parts)


# ---- textures ---------------------------------------------------------------------

def checker_texture(size, seed, c0=(200, 180, 150), c1=(90, 70, 60), cells=8):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:size, 0:size]
    m = (((x * cells) // size + (y * cells) // size) & 1).astype(bool)
    img = np.where(m[..., None], np.array(c0)[None, None], np.array(c1)[None, None]).astype(np.int32)
    img = img + rng.integers(-12, 13, size=(size, size, 1))
    return np.clip(img, 0, 255).astype(np.uint8)


def noise_texture(size, seed, base=(170, 160, 150), amp=60):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:size, 0:size].astype(np.float64) / size
    acc = np.zeros((size, size))
    for k in range(1, 6):
        fx, fy = rng.integers(1, 9, 2) * k
        acc += np.sin(2 * np.pi * (fx * x + fy * y) + rng.uniform(0, 6.28)) / k
    acc = acc / np.abs(acc).max()
    img = np.array(base)[None, None, :] + amp * acc[..., None] * np.array([1.0, 0.9, 0.8])[None, None, :]
    return np.clip(img, 0, 255).astype(np.uint8)


def sky_faces(size=512):
    """Six gradient faces in the reference's order front(+Z), back(-Z), left(-X), right(+X), top(+Y), bottom(-Y)."""
    v = np.linspace(0, 1, size)[:, None, None]   # row 0 = v 0 = horizon-down side of a wall face
    u = np.linspace(0, 1, size)[None, :, None]
    horizon = np.array([235, 225, 205.0])
    zenith = np.array([90, 140, 230.0])
    faces = []
    for k in range(4):
        g = horizon * (1 - v) + zenith * v + 12 * np.sin(2 * np.pi * (u * (k + 1)))
        faces.append(np.clip(g + 0 * u, 0, 255).astype(np.uint8))
    rr = np.sqrt((u - 0.5) ** 2 + (v - 0.5) ** 2)
    top = zenith[None, None, :] * (1 - 0.3 * rr) + 255 * np.exp(-((rr / 0.08) ** 2))
    faces.append(np.clip(top, 0, 255).astype(np.uint8))
    faces.append(np.clip(np.full((size, size, 3), 60.0), 0, 255).astype(np.uint8))
    return faces


# ---- scenes -----------------------------------------------------------------------

class Scene:
    """Everything the C ABI needs: triangles, textures, sky, camera look-at and render parameters."""

    def __init__(self, name, tris, textures, sky, eye, at, focal, rotate=0.0, width=64, height=64, max_depth=2,
                 n_diff=1, n_spec=1, bg=(0.0, 0.0, 0.0), seed=0x5EED):
        self.name, self.tris, self.textures, self.sky = name, tris, textures, sky
        self.eye, self.at, self.focal, self.rotate = eye, at, focal, rotate
        self.width, self.height, self.max_depth, self.n_diff, self.n_spec = width, height, max_depth, n_diff, n_spec
        self.bg, self.seed = bg, seed


def mixed_scene(seed=1, width=64, height=64, max_depth=3, n_diff=2, n_spec=1, with_sky=True, ior=1.5,
                with_tex=True, detail=1.0):
    """Small scene with every material class, a textured part, coplanar duplicates, axis-aligned flat
    triangles, a light, glass, and (optionally) a cube-map sky. ~1.5k triangles at detail 1."""
    rng = np.random.default_rng(seed)
    sb = SceneBuilder()
    tex0 = sb.add_texture(checker_texture(16, seed)) if with_tex else -1
    tex1 = sb.add_texture(noise_texture(32, seed + 1)) if with_tex else -1
    if with_sky:
        sb.set_skybox(0.9, sky_faces(32))
    d = lambda k: max(2, int(round(k * detail)))
    # floor (axis-aligned, flat boxes in the BVH), textured dielectric
    sb.grid((-40, 0, -40), (0, 0, 80), (80, 0, 0), d(6), d(6), Material((0.9, 0.9, 0.9), rough=0.3, tex=tex0))
    # back wall, plain dielectric
    sb.grid((-40, 0, -40), (80, 0, 0), (0, 50, 0), d(3), d(3), Material((0.7, 0.75, 0.9), rough=0.8))
    # light
    sb.grid((-12, 45, -12), (24, 0, 0), (0, 0, 24), 1, 1, Material((6.0, 5.5, 5.0), emit=True))
    # bumpy metal sphere
    ph = rng.uniform(0, 6.28, 3)
    sb.sphere((-14, 10, 0), 9.0, d(12), d(16),
              Material((0.95, 0.8, 0.5), metal=True, rough=0.15),
              disp=lambda t, p: 0.08 * np.sin(3 * t + ph[0]) * np.cos(4 * p + ph[1]))
    # glass sphere
    sb.sphere((12, 9, 6), 8.0, d(10), d(14), Material((1, 1, 1), transp=True, rough=0.02, ior=ior))
    # textured dielectric blob
    sb.sphere((0, 7, -16), 7.0, d(10), d(14), Material((0.9, 0.6, 0.5), rough=0.6, tex=tex1),
              disp=lambda t, p: 0.15 * np.sin(5 * p + ph[2]) * np.sin(2 * t))
    # a box
    sb.box((18, 0, -22), (30, 14, -10), Material((0.4, 0.8, 0.4), rough=0.5), n=d(2))
    tris = sb.triangles()
    # coplanar exact duplicates (equal t ties) and a sliver
    dup = tris[:4].copy()
    dup["color"][:, :3] = (1.0, 0.2, 0.2)
    tris = np.concatenate([tris, dup])
    return Scene("mixed", tris, sb.textures, sb.sky, eye=(5.0, 24.0, 62.0), at=(0.0, 8.0, 0.0), focal=30.0,
                 width=width, height=height, max_depth=max_depth, n_diff=n_diff, n_spec=n_spec,
                 bg=(0.05, 0.06, 0.08), seed=seed)


def bunny_class_scene(width=1920, height=1080, rings=186, segs=187, seed=7):
    """Closed genus-0 displaced sphere, rings*segs*2 = 69,564 triangles (Stanford-bunny class);
    primary rays only matter for this workload."""
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 6.28, 8)
    sb = SceneBuilder()

    def disp(t, p):
        return (0.18 * np.sin(2 * t + ph[0]) * np.cos(3 * p + ph[1]) + 0.08 * np.sin(7 * t + ph[2]) * np.sin(5 * p + ph[3])
                + 0.03 * np.sin(17 * t + ph[4]) * np.cos(13 * p + ph[5]) + 0.012 * np.sin(41 * t + ph[6]) * np.sin(37 * p + ph[7]))

    sb.sphere((0, 0, 0), 10.0, rings, segs, Material((0.8, 0.8, 0.8), rough=0.5), disp=disp)
    tris = sb.triangles()
    return Scene("bunny_class", tris, [], None, eye=(6.0, 9.0, 30.0), at=(0.0, 0.0, 0.0), focal=38.0,
                 width=width, height=height, max_depth=1, n_diff=1, n_spec=1, bg=(0.2, 0.3, 0.5), seed=seed)


def atrium_scene(width=1920, height=1080, max_depth=4, n_diff=1, n_spec=1, detail=1.0, tex_size=512, seed=11):
    """Sponza-class atrium: tessellated floor and walls, two storeys of fluted columns and arches, hanging
    cloth strips, a glass sphere and a metal sphere, one emissive quad, per-part textures and a cube-map
    sky visible through the open roof. About 262k triangles at detail 1."""
    rng = np.random.default_rng(seed)
    sb = SceneBuilder()
    sb.set_skybox(2.5, sky_faces(tex_size))
    t_floor = sb.add_texture(checker_texture(tex_size, seed, cells=16))
    t_wall = sb.add_texture(noise_texture(tex_size, seed + 1))
    t_col = sb.add_texture(noise_texture(tex_size, seed + 2, base=(200, 190, 170), amp=35))
    t_cloth = [sb.add_texture(noise_texture(tex_size, seed + 3 + k, base=b, amp=50))
               for k, b in enumerate([(190, 40, 40), (40, 80, 180), (40, 150, 70)])]
    d = lambda k: max(2, int(round(k * detail)))
    L, Wd, H = 120.0, 50.0, 60.0   # x extent, z extent, height
    bump = lambda s, t: 0.15 * np.sin(40 * s) * np.sin(40 * t)
    # floor and the two storey floors
    sb.grid((-L / 2, 0, -Wd / 2), (0, 0, Wd), (L, 0, 0), d(90), d(150), Material((0.9, 0.9, 0.9), rough=0.25, tex=t_floor), disp=bump)
    # long walls (facing inward) and end walls
    sb.grid((-L / 2, 0, -Wd / 2), (L, 0, 0), (0, H, 0), d(120), d(60), Material((0.85, 0.8, 0.7), rough=0.9, tex=t_wall), disp=bump)
    sb.grid((L / 2, 0, Wd / 2), (-L, 0, 0), (0, H, 0), d(120), d(60), Material((0.85, 0.8, 0.7), rough=0.9, tex=t_wall), disp=bump)
    sb.grid((-L / 2, 0, Wd / 2), (0, 0, -Wd), (0, H, 0), d(50), d(60), Material((0.8, 0.8, 0.75), rough=0.9, tex=t_wall), disp=bump)
    sb.grid((L / 2, 0, -Wd / 2), (0, 0, Wd), (0, H, 0), d(50), d(60), Material((0.8, 0.8, 0.75), rough=0.9, tex=t_wall), disp=bump)
    # colonnades: two rows, two storeys
    ncol = 9
    for storey in range(2):
        y0 = storey * 26.0
        for row, z in enumerate((-Wd / 2 + 10, Wd / 2 - 10)):
            xs = np.linspace(-L / 2 + 10, L / 2 - 10, ncol)
            for k, x in enumerate(xs):
                sb.cylinder((x, y0, z), 1.8, 20.0, d(32), d(32), Material((0.9, 0.88, 0.8), rough=0.7, tex=t_col), flute=0.06)
                sb.box((x - 2.6, y0 + 20.0, z - 2.6), (x + 2.6, y0 + 22.0, z + 2.6), Material((0.85, 0.85, 0.8), rough=0.8), n=d(3))
                if k + 1 < ncol:
                    xm = 0.5 * (x + xs[k + 1])
                    sb.arch((xm, y0 + 20.0, z), 0, 0.5 * (xs[k + 1] - x) - 1.0, 1.2, d(36), d(16), Material((0.88, 0.85, 0.8), rough=0.8, tex=t_col))
            # gallery slab behind each row
            zs = z - 10 if z < 0 else z
            sb.box((-L / 2, y0 + 24.0, zs), (L / 2, y0 + 26.0, zs + 10), Material((0.8, 0.8, 0.8), rough=0.9), n=d(10))
    # hanging cloth strips
    for k in range(6):
        x = -45.0 + 18.0 * k
        phase = rng.uniform(0, 6.28)
        wave = lambda s, t, phase=phase: 1.6 * np.sin(6.0 * s + phase) * (0.3 + t) + 0.5 * np.sin(19 * t + phase)
        sb.grid((x, 50.0, -6.0), (0, -34.0, 0), (0, 0, 12.0), d(80), d(30),
                Material((0.9, 0.9, 0.9), rough=1.0, tex=t_cloth[k % 3]), disp=wave)
        sb.grid((x, 50.0, 6.0), (0, -34.0, 0), (0, 0, -12.0), d(80), d(30),
                Material((0.9, 0.9, 0.9), rough=1.0, tex=t_cloth[k % 3]), disp=lambda s, t, w=wave: -w(s, 1 - t))
    # feature objects
    sb.sphere((-20, 7, 0), 7.0, d(60), d(80), Material((0.95, 0.85, 0.6), metal=True, rough=0.05))
    sb.sphere((18, 6, 4), 6.0, d(60), d(80), Material((1, 1, 1), transp=True, rough=0.01, ior=1.5))
    # emissive quad under the upper gallery
    sb.grid((-25, 49.5, -8), (50, 0, 0), (0, 0, 16), 1, 1, Material((14.0, 13.0, 11.0), emit=True))
    tris = sb.triangles()
    return Scene("atrium", tris, sb.textures, sb.sky, eye=(-52.0, 14.0, 3.0), at=(30.0, 16.0, -2.0), focal=22.0,
                 width=width, height=height, max_depth=max_depth, n_diff=n_diff, n_spec=n_spec,
                 bg=(0.0, 0.0, 0.0), seed=seed)


def heightfield_scene(n_tris=2048, width=512, height=512, seed=3):
    """The reference's template scene (config.txt:4-15,51-61: camera 50 50 0 -> origin, focal 400... scaled),
    its floor triangle, plus a generated ~2k-triangle bumpy mesh standing in for the placeholder model."""
    side = int(round(np.sqrt(n_tris / 2)))
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 6.28, 4)
    sb = SceneBuilder()
    bump = lambda s, t: 6.0 * np.sin(5 * s + ph[0]) * np.cos(4 * t + ph[1]) + 2.0 * np.sin(13 * s + ph[2]) * np.sin(11 * t + ph[3])
    # y points down in the template (floor at y=100 faces -y): the bumps rise towards the camera
    sb.grid((-60, 60, -60), (120, 0, 0), (0, 0, 120), side, side, Material((0.8, 0.8, 0.8), metal=True, rough=1.0), disp=bump)
    sb.grid((-30, -40, -30), (60, 0, 0), (0, 0, 60), 1, 1, Material((8, 8, 8), emit=True), flip=True)
    tris = sb.triangles()
    return Scene("template", tris, [], None, eye=(250.0, -250.0, 0.0), at=(0.0, 60.0, 0.0), focal=60.0,
                 width=width, height=height, max_depth=2, n_diff=2, n_spec=2, bg=(0.1, 0.1, 0.1), seed=seed)
