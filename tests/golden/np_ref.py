"""A THIRD restatement of one bounce of the reference, in numpy float32 scalars — no C++, no HIP, none of the
product or oracle code.  It exists to pin the oracle (oracle/oracle.cpp) and the GPU kernels against something that
was written separately from both: tests/golden/make_golden.py evaluates it on fixed inputs and commits the results
as tests/golden/bounce_vectors.json; tests/test_golden_vectors.py checks the oracle (CPU) and rt_debug_bounce (GPU,
both the single-kernel hook and the production kernels) against those vectors.

Everything follows /root/reference/src by file:line; arithmetic is IEEE binary32 in glam 0.21's operation order
(dot = (x*x' + y*y') + z*z', normalize = v * (1 / length)), one np.float32 operation per reference operation.  The
reference itself ships no vectors, so this is still "parity unpinned" — but it is a third reading.

Random numbers: the counter generator of DESIGN.md 3 (the reference's thread-local SmallRng cannot be reproduced by a
one-ray-per-lane renderer), drawn in the reference's order."""
import math

import numpy as np

f32 = np.float32
U32 = 0xFFFFFFFF
PI = f32(math.pi)
FRAC_1_PI = f32(1.0 / math.pi)


# ---------------------------------------------------------------------------------------------------------------
# glam Vec3A
# ---------------------------------------------------------------------------------------------------------------
def v3(x, y, z):
    return (f32(x), f32(y), f32(z))


def add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def mul(a, b): return (a[0] * b[0], a[1] * b[1], a[2] * b[2])
def scale(a, s): return (a[0] * s, a[1] * s, a[2] * s)
def divs(a, s): return (a[0] / s, a[1] / s, a[2] / s)
def neg(a): return (-a[0], -a[1], -a[2])
def dot(a, b): return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]
def length(a): return np.sqrt(dot(a, a))
def normalize(a): return scale(a, f32(1.0) / length(a))
def cross(a, b): return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
def lerp3(a, b, s): return add(a, scale(sub(b, a), s))
def lerp(a, b, s): return a + (b - a) * s  # math.rs:154-156


# ---------------------------------------------------------------------------------------------------------------
# counter RNG: draw(k0, k1, ctr) = mix32((k0 ^ ctr * 0x9E3779B9) + k1); f32 = (u32 >> 8) * 2^-24
# ---------------------------------------------------------------------------------------------------------------
def fmix32(h):
    h &= U32
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & U32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & U32
    h ^= h >> 16
    return h


def mix32(h):  # the per-draw finaliser (lowbias32)
    h &= U32
    h ^= h >> 16
    h = (h * 0x7FEB352D) & U32
    h ^= h >> 15
    h = (h * 0x846CA68B) & U32
    h ^= h >> 16
    return h


def path_key(seed, pix, samp):
    s_lo, s_hi = seed & U32, (seed >> 32) & U32
    a = fmix32(pix ^ s_lo)
    return fmix32((a + samp * 0x9E3779B9 + s_hi) & U32), fmix32(((a ^ 0xA511E9B3) + samp * 0xC2B2AE3D) & U32)


class Rng:
    def __init__(self, k0, k1, depth):
        self.k0, self.k1, self.ctr = int(k0), int(k1), (depth + 1) * 256

    def next(self):  # rand's Standard f32: top 24 bits of a u32 times 2^-24 (main.rs:89-90, math.rs:19-21)
        r = mix32(((self.k0 ^ ((self.ctr * 0x9E3779B9) & U32)) + self.k1) & U32)
        self.ctr += 1
        return f32(r >> 8) * f32(1.0 / 16777216.0)


# ---------------------------------------------------------------------------------------------------------------
# math.rs
# ---------------------------------------------------------------------------------------------------------------
def random_in_unit_sphere(rng):  # math.rs:17-37
    while True:
        x, y, z = rng.next(), rng.next(), rng.next()
        v = (x * (f32(1.0) - f32(-1.0)) + f32(-1.0), y * (f32(1.0) - f32(-1.0)) + f32(-1.0), z * (f32(1.0) - f32(-1.0)) + f32(-1.0))
        if dot(v, v) < f32(1.0):
            return v


def random_on_hemisphere(rng, n):  # math.rs:43-53
    v = random_in_unit_sphere(rng)
    if not (dot(v, n) > f32(0.0)):
        v = neg(v)
    return normalize(v)


def reflect(v, n):  # math.rs:68-70
    return sub(v, scale(n, f32(2.0) * dot(v, n)))


def refract(uv, n, eta):  # math.rs:72-77 (the method call binds before the unary minus)
    cos_theta = -min(dot(uv, n), f32(1.0))
    perp = scale(add(uv, scale(n, cos_theta)), eta)
    par = scale(n, -np.sqrt(abs(f32(1.0) - dot(perp, perp))))
    return add(perp, par)


def powi5(x):  # LLVM's powi(5): x * ((x*x) * (x*x))
    x2 = x * x
    return x * (x2 * x2)


def schlick_fresnel(u): return powi5(f32(1.0) - u)  # math.rs:79-81


def reflectance(cosine, ref_idx):  # math.rs:84-88
    r0 = (f32(1.0) - ref_idx) / (f32(1.0) + ref_idx)
    r0 = r0 * r0
    return r0 + (f32(1.0) - r0) * schlick_fresnel(cosine)


def _as_i32_sat(x):  # Rust `as i32`
    if x != x:
        return 0
    return int(max(-2147483648.0, min(2147483647.0, math.trunc(float(x)))))


def offset_hit_point(p, n):  # math.rs:137-152
    out = []
    for pc, nc in zip(p, n):
        of_i = _as_i32_sat(nc * f32(256.0))
        bits = int(np.array([pc], dtype=np.float32).view(np.uint32)[0])
        bits = (bits + (-of_i if pc < f32(0.0) else of_i)) & U32  # i32 wrapping add == u32 add of the bit pattern
        p_i = np.array([bits], dtype=np.uint32).view(np.float32)[0]
        out.append(pc + nc * f32(1.0 / 65536.0) if abs(pc) < f32(1.0 / 32.0) else p_i)
    return tuple(out)


# ---------------------------------------------------------------------------------------------------------------
# hitable.rs
# ---------------------------------------------------------------------------------------------------------------
def get_uv(n):  # hitable.rs:65-71
    theta = f32(np.arccos(-n[1]))
    phi = f32(np.arctan2(-n[2], n[0])) + PI
    return phi / (f32(2.0) * PI), theta / PI


def sphere_hit(c, r, o, d, t_min, t_max):  # hitable.rs:75-102
    oc = sub(o, c)
    a = dot(d, d)
    half_b = dot(oc, d)
    cc = dot(oc, oc) - r * r
    disc = half_b * half_b - a * cc
    if disc < f32(0.0):
        return None
    sqrtd = np.sqrt(disc)
    root = (-half_b - sqrtd) / a
    if root < t_min or t_max < root:
        root = (-half_b + sqrtd) / a
        if root < t_min or t_max < root:
            return None
    p = add(o, scale(d, root))
    on = divs(sub(p, c), r)
    front = dot(d, on) < f32(0.0)  # hitable.rs:25-32
    return {"t": root, "p": p, "on": on, "front": front, "n": on if front else neg(on)}


def rotate_y_sphere_hit(sin_t, cos_t, c, r, o, d, t_min, t_max):
    """RotateY::hit over a Sphere (hitable.rs:469-511): the ray is rotated into object space, the record's point and
    (already face-flipped) normal are rotated back, and set_face_normal runs a second time with the OBJECT-space
    direction against the WORLD-space normal (hitable.rs:505) — the quirk the flat scene reproduces."""
    ro = (cos_t * o[0] - sin_t * o[2], o[1], sin_t * o[0] + cos_t * o[2])
    rd = (cos_t * d[0] - sin_t * d[2], d[1], sin_t * d[0] + cos_t * d[2])
    h = sphere_hit(c, r, ro, rd, t_min, t_max)
    if h is None:
        return None
    p, n = h["p"], h["n"]
    p2 = (cos_t * p[0] + sin_t * p[2], p[1], -sin_t * p[0] + cos_t * p[2])
    n2 = (cos_t * n[0] + sin_t * n[2], n[1], -sin_t * n[0] + cos_t * n[2])
    front = dot(rd, n2) < f32(0.0)
    h.update(p=p2, front=front, n=n2 if front else neg(n2))
    return h


def rect_hit(axis, mn, mx, o, d, t_min, t_max):
    """XYRect / XZRect / YZRect::hit (hitable.rs:251-270, 291-310, 331-350): axis = the constant coordinate (0 x, 1 y, 2 z),
    the plane is min[axis]; uv = ((p - min) / (max - min)) over the two other axes; outward normal = +axis."""
    t = (mn[axis] - o[axis]) / d[axis]
    if t < t_min or t > t_max:
        return None
    p = add(o, scale(d, t))
    ua, va = ((1, 2), (0, 2), (0, 1))[axis]
    if p[ua] < mn[ua] or p[ua] > mx[ua] or p[va] < mn[va] or p[va] > mx[va]:
        return None
    on = tuple(f32(1.0) if k == axis else f32(0.0) for k in range(3))
    front = dot(d, on) < f32(0.0)
    return {"t": t, "p": p, "on": on, "front": front, "n": on if front else neg(on),
            "uv": ((p[ua] - mn[ua]) / (mx[ua] - mn[ua]), (p[va] - mn[va]) / (mx[va] - mn[va]))}


def translate_hit(offset, inner_hit, o, d, t_min, t_max):  # Translate::hit hitable.rs:409-418
    h = inner_hit(sub(o, offset), d, t_min, t_max)
    if h is not None:
        h["p"] = add(h["p"], offset)
    return h


def constant_medium_hit(boundary_hit, neg_inv_density, o, d, t_min, t_max, xi):
    """ConstantMedium::hit (hitable.rs:536-579) over any boundary; xi = the uniform draw of hitable.rs:564."""
    inf = f32(np.inf)
    r1 = boundary_hit(o, d, -inf, inf)
    if r1 is None:
        return None
    r2 = boundary_hit(o, d, r1["t"] + f32(0.0001), inf)
    if r2 is None:
        return None
    t1, t2 = r1["t"], r2["t"]
    if t1 < t_min:
        t1 = t_min
    if t2 > t_max:
        t2 = t_max
    if t1 >= t2:
        return None
    if t1 < f32(0.0):
        t1 = f32(0.0)
    ray_len = length(d)
    dist_inside = (t2 - t1) * ray_len
    hit_dist = neg_inv_density * f32(np.log(xi))
    if hit_dist > dist_inside:
        return None
    t = t1 + hit_dist / ray_len
    nx = v3(1, 0, 0)  # hitable.rs:574-575: rec.norm = Vec3A::X, front_face = true
    return {"t": t, "p": add(o, scale(d, t)), "on": nx, "front": True, "n": nx}


# ---------------------------------------------------------------------------------------------------------------
# texture.rs
# ---------------------------------------------------------------------------------------------------------------
def perlin_noise(vec, perm, p):  # texture.rs:125-146, 93-112; vec [256][3] f32, perm [3][256]
    fl = (np.floor(p[0]), np.floor(p[1]), np.floor(p[2]))
    ijk = (int(fl[0]), int(fl[1]), int(fl[2]))
    uvw = sub(p, fl)
    s = tuple(x * x * (f32(3.0) - f32(2.0) * x) for x in uvw)  # math.rs:133-135
    accum = f32(0.0)
    for a in range(2):
        for b in range(2):
            for c in range(2):
                idx = perm[0][(ijk[0] + a) % 256] ^ perm[1][(ijk[1] + b) % 256] ^ perm[2][(ijk[2] + c) % 256]  # rem_euclid
                g = (f32(vec[idx][0]), f32(vec[idx][1]), f32(vec[idx][2]))
                w = sub(uvw, v3(a, b, c))
                accum = accum + dot(g, w) * (s[0] if a else f32(1.0) - s[0]) * (s[1] if b else f32(1.0) - s[1]) * (
                    s[2] if c else f32(1.0) - s[2])
    return accum


def perlin_turb(vec, perm, p):  # texture.rs:115-124
    accum, w = f32(0.0), f32(1.0)
    for _ in range(7):
        accum = accum + w * perlin_noise(vec, perm, p)
        p = scale(p, f32(2.0))
        w = w * f32(0.5)
    return abs(accum)


def perlin_value(vec, perm, tex_scale, p):  # texture.rs:164-168
    s = f32(np.sin(f32(10.0) * perlin_turb(vec, perm, p) + tex_scale * p[2]))
    g = (s + f32(1.0)) * f32(0.5)
    return (g, g, g)


def _as_u32_sat(x):  # Rust `as u32`: saturating, NaN -> 0
    if x != x or x <= 0:
        return 0
    return int(min(4294967295.0, math.trunc(float(x))))


def image_value(img, uv):  # texture.rs:183-193; img [h][w][3] f32, row 0 = top; Rust f32::clamp keeps NaN
    h, w = img.shape[0], img.shape[1]

    def clamp01(x):
        if x < f32(0.0):
            return f32(0.0)
        if x > f32(1.0):
            return f32(1.0)
        return x
    u = clamp01(uv[0])
    v = f32(1.0) - clamp01(uv[1])
    i = min(_as_u32_sat(u * f32(w)), w - 1)
    j = min(_as_u32_sat(v * f32(h)), h - 1)
    return tuple(f32(x) for x in img[j][i])


# ---------------------------------------------------------------------------------------------------------------
# pbr.rs helpers
# ---------------------------------------------------------------------------------------------------------------
def gtr1(n_dot_h, a):  # pbr.rs:72-79
    if a >= f32(1.0):
        return FRAC_1_PI
    a2 = a * a
    t = f32(1.0) + (a2 - f32(1.0)) * n_dot_h * n_dot_h
    return (a2 - f32(1.0)) / (PI * f32(np.log(a2)) * t)


def gtr2(n_dot_h, a):  # pbr.rs:81-85
    a2 = a * a
    t = f32(1.0) + (a2 - f32(1.0)) * n_dot_h * n_dot_h
    return a2 / (PI * t * t)


def gtr2_aniso(h, ax, ay):  # pbr.rs:87-89
    hx, hy = h[0] / ax, h[1] / ay
    q = hx * hx + hy * hy + h[2] * h[2]
    return f32(1.0) / (PI * ax * ay * (q * q))


def smith_geo_ggx(n_dot_v, alpha):  # pbr.rs:92-96
    a, b = alpha * alpha, n_dot_v * n_dot_v
    return f32(1.0) / (n_dot_v + np.sqrt(a + b - a * b))


def smith_geo_ggx_aniso(v, ax, ay):  # pbr.rs:98-100
    x, y = v[0] * ax, v[1] * ay
    return f32(1.0) / (v[2] + np.sqrt(x * x + y * y + v[2] * v[2]))


def fresnel_dielectric(n_dot_i, n_dot_t, eta):  # pbr.rs:107-113
    rs = (n_dot_i - eta * n_dot_t) / (n_dot_i + eta * n_dot_t)
    rp = (eta * n_dot_i - n_dot_t) / (eta * n_dot_i + n_dot_t)
    return (rs * rs + rp * rp) / f32(2.0)


def fresnel_dielectric_2(n_dot_i, eta):  # pbr.rs:120-129
    n_dot_t_sq = f32(1.0) - (f32(1.0) - n_dot_i * n_dot_i) / (eta * eta)
    if n_dot_t_sq < f32(0.0):
        return f32(1.0)
    return fresnel_dielectric(abs(n_dot_i), np.sqrt(n_dot_t_sq), eta)


def smith_masking_gtr2_2(v, n, roughness):  # pbr.rs:145-152
    alpha = roughness * roughness
    a2 = alpha * alpha
    z = dot(v, n)
    z2 = z * z
    lam = (f32(-1.0) + np.sqrt(f32(1.0) + a2 * (f32(1.0) - z2) / z2)) / f32(2.0)
    return f32(1.0) / (f32(1.0) + lam)


def world_to_local_with_rot(norm, tang0, v, rot):  # hitable.rs:37-41
    tang = sub(scale(tang0, f32(np.cos(rot))), scale(cross(norm, tang0), f32(np.sin(rot))))
    bitang = cross(norm, tang)
    return (dot(v, tang), dot(v, bitang), dot(v, norm))


# ---------------------------------------------------------------------------------------------------------------
# material.rs / pbr.rs scatter.  `mat` = dict(type=RtMatType, tex=(r,g,b) value of the first texture at the hit,
# tex1=..., color=..., p0..p2).  Returns (alive, attenuation, scattered_o, scattered_d, emitted)
# ---------------------------------------------------------------------------------------------------------------
ZERO = v3(0, 0, 0)


def scatter(mat, rd, rec, rng):
    ty = mat["type"]
    n, p = rec["n"], rec["p"]
    tex = tuple(f32(x) for x in mat.get("tex", (0, 0, 0)))
    p0, p1, p2 = f32(mat.get("p0", 0)), f32(mat.get("p1", 0)), f32(mat.get("p2", 0))
    if ty == 0:  # Emission material.rs:21-28
        return False, ZERO, ZERO, ZERO, tex
    if ty == 1:  # Diffuse material.rs:35-46
        sd = add(n, normalize(random_in_unit_sphere(rng)))
        eps = f32(np.finfo(np.float32).eps)
        if abs(sd[0]) < eps and abs(sd[1]) < eps and abs(sd[2]) < eps:  # math.rs:8-11
            sd = n
        return True, tex, offset_hit_point(p, n), normalize(sd), ZERO
    if ty == 3:  # Metal material.rs:66-73 (the ball is drawn even when fuzz == 0)
        refl = add(reflect(rd, n), scale(random_in_unit_sphere(rng), p0))
        col = tuple(f32(x) for x in mat["color"])
        return bool(dot(refl, n) > f32(0.0)), col, p, normalize(refl), ZERO
    if ty == 4:  # Dielectric material.rs:79-97
        ref_idx = f32(1.0) / p0 if rec["front"] else p0
        cos_theta = -min(dot(rd, n), f32(1.0))
        sin_theta = np.sqrt(f32(1.0) - cos_theta * cos_theta)
        cannot = sin_theta * ref_idx > f32(1.0)
        xi = rng.next()
        d = reflect(rd, n) if (cannot or reflectance(cos_theta, ref_idx) > xi) else refract(rd, n, ref_idx)
        return True, v3(1, 1, 1), p, normalize(d), ZERO
    if ty == 5:  # Isotropic material.rs:103-113
        return True, tex, p, normalize(random_in_unit_sphere(rng)), ZERO
    # Lambert and the seven pbr.rs materials: offset origin + uniform hemisphere direction
    po = offset_hit_point(p, n)
    dir_o = random_on_hemisphere(rng, n)
    n_dot_i = dot(n, neg(rd))
    n_dot_o = dot(n, dir_o)
    two = f32(2.0)
    if ty == 2:  # Lambert material.rs:52-59
        att = scale(scale(tex, two), dot(n, dir_o))
    elif ty == 6:  # OrenNayar pbr.rs:17-42
        cos_i, cos_o = abs(dot(n, rd)), n_dot_o
        sin_i, sin_o = np.sqrt(f32(1.0) - cos_i * cos_i), np.sqrt(f32(1.0) - cos_o * cos_o)
        max_cos = max(cos_i * cos_o + sin_i * sin_o, f32(0.0))
        r2 = p0 * p0
        a = f32(1.0) - f32(0.5) * r2 / (r2 + f32(0.33))
        b = f32(0.45) * r2 / (r2 + f32(0.09))
        sin_alpha, tan_beta = (sin_o, sin_i / cos_i) if cos_i > cos_o else (sin_i, sin_o / cos_o)
        w = a + b * max_cos * sin_alpha * tan_beta
        att = scale(scale(scale(tex, w), two), cos_o)
    elif ty == 7:  # BurleyDiffuse pbr.rs:50-69
        h = normalize(sub(dir_o, rd))
        h_dot_o = dot(h, dir_o)
        fl, fv = schlick_fresnel(n_dot_o), schlick_fresnel(n_dot_i)
        fd90 = f32(0.5) + two * h_dot_o * h_dot_o * p0
        fd = lerp(f32(1.0), fd90, fl) * lerp(f32(1.0), fd90, fv)
        att = scale(scale(scale(tex, fd), two), n_dot_o)
    elif ty == 8:  # RoughPlastic pbr.rs:160-189 (tex = spec_color, tex1 = diff_color)
        h = normalize(sub(dir_o, rd))
        h_dot_i, h_dot_o, n_dot_h = dot(h, neg(rd)), dot(h, dir_o), dot(n, h)
        kd = tuple(f32(x) for x in mat["tex1"])
        rough = min(max(p0, f32(0.01)), f32(1.0))
        f_o = fresnel_dielectric_2(h_dot_o, p1)
        dd = gtr2(n_dot_h, rough)
        gg = smith_masking_gtr2_2(neg(rd), n, rough) * smith_masking_gtr2_2(dir_o, n, rough)
        spec = divs(scale(tex, gg * f_o * dd), f32(4.0) * n_dot_i * n_dot_o)
        f_i = fresnel_dielectric_2(h_dot_i, p1)
        diff = scale(scale(scale(kd, f32(1.0) - f_o), f32(1.0) - f_i), FRAC_1_PI)
        att = scale(scale(scale(add(spec, diff), n_dot_o), two), PI)
    elif ty == 9:  # DisneyDiffuse pbr.rs:198-222
        h = normalize(sub(dir_o, rd))
        h_dot_o = dot(h, dir_o)
        fo, fi = schlick_fresnel(n_dot_o), schlick_fresnel(n_dot_i)
        fd90 = f32(0.5) + two * h_dot_o * h_dot_o * p0
        fd = lerp(f32(1.0), fd90, fo) * lerp(f32(1.0), fd90, fi)
        fss90 = p0 * h_dot_o * h_dot_o
        fss = f32(1.25) * (lerp(f32(1.0), fss90, fi) * lerp(f32(1.0), fss90, fo) * (f32(1.0) / (n_dot_i + n_dot_o) - f32(0.5)) + f32(0.5))
        att = scale(scale(scale(tex, lerp(fd, fss, p1)), two), n_dot_o)
    elif ty == 10:  # DisneyMetal pbr.rs:232-278 (p0 roughness, p1 anisotropic, p2 rot)
        h = normalize(sub(dir_o, rd))
        h_dot_o, n_dot_h = dot(h, dir_o), dot(n, h)
        fm = lerp3(tex, v3(1, 1, 1), schlick_fresnel(h_dot_o))
        alpha_min = f32(0.0001)
        if p1 > f32(-10.0):
            aspect = np.sqrt(f32(1.0) - f32(0.9) * p1)
            ax, ay = max(p0 * p0 / aspect, alpha_min), max(p0 * p0 * aspect, alpha_min)
            rot = p2 * two * PI
            tang = normalize(cross(v3(0, 1, 0), rec["on"]))  # hitable.rs:96: from the OUTWARD normal
            dm = gtr2_aniso(world_to_local_with_rot(n, tang, h, rot), ax, ay)
            gm = smith_geo_ggx_aniso(world_to_local_with_rot(n, tang, neg(rd), rot), ax, ay) * smith_geo_ggx_aniso(
                world_to_local_with_rot(n, tang, dir_o, rot), ax, ay)
        else:
            r2 = max(p0 * p0, alpha_min)
            dm = gtr2(n_dot_h, r2)
            gm = smith_geo_ggx(n_dot_i, r2) * smith_geo_ggx(n_dot_o, r2)
        att = scale(scale(scale(scale(scale(fm, dm), gm), n_dot_o), two), PI)
    elif ty == 11:  # DisneySheen pbr.rs:286-308
        h = normalize(sub(dir_o, rd))
        h_dot_o = dot(h, dir_o)
        lum = dot(v3(0.3, 0.6, 0.1), tex)
        c_tint = divs(tex, lum) if lum > f32(0.0) else v3(1, 1, 1)
        c_sheen = lerp3(v3(1, 1, 1), c_tint, p0)
        att = scale(scale(scale(scale(c_sheen, schlick_fresnel(h_dot_o)), n_dot_o), two), PI)
    elif ty == 12:  # DisneyClearcoat pbr.rs:314-335
        h = normalize(sub(dir_o, rd))
        h_dot_o, n_dot_h = dot(h, dir_o), dot(n, h)
        fc = lerp(f32(0.4), f32(1.0), schlick_fresnel(h_dot_o))
        dc = gtr1(n_dot_h, lerp(f32(0.1), f32(0.001), p0))
        gc = smith_geo_ggx(n_dot_i, f32(0.25)) * smith_geo_ggx(n_dot_o, f32(0.25))
        cc = f32(0.25) * fc * dc * gc
        att = scale(scale(scale((cc, cc, cc), n_dot_o), two), PI)
    else:
        raise ValueError(ty)
    return True, att, po, dir_o, ZERO


def sky_gradient(d):  # demo_scene.rs:28-31
    t = d[1] * f32(0.5) + f32(0.5)
    return lerp3(v3(1, 1, 1), v3(0.5, 0.7, 1.0), t)


def bounce(case, o, d, key, depth):
    """main.rs:44-58 for one segment against ONE sphere (optionally under a RotateY wrapper), gradient sky.
    case: dict(c, r, mat, rot=(sin, cos) or None, tex_eval=callable(rec) -> rgb or None)."""
    tmax = f32(np.finfo(np.float32).max)
    if case.get("rect"):  # an axis-aligned rectangle below a Translate
        rc = case["rect"]
        mn, mx, off = v3(*rc["min"]), v3(*rc["max"]), v3(*rc["offset"])
        h = translate_hit(off, lambda ro, rd, a, b: rect_hit(rc["axis"], mn, mx, ro, rd, a, b), o, d, f32(1e-3), tmax)
        return _finish(case, h, d, key, depth)
    c, r = v3(*case["c"]), f32(case["r"])
    if case.get("medium"):  # ConstantMedium over the sphere; its free-path draw is counter slot 224 + medium index
        g = Rng(key[0], key[1], depth)
        g.ctr += 224
        xi = g.next()
        h = constant_medium_hit(lambda ro, rd, a, b: sphere_hit(c, r, ro, rd, a, b), f32(case["medium"]["neg_inv_density"]),
                                o, d, f32(1e-3), tmax, xi)
        return _finish(case, h, d, key, depth)
    if case.get("rot"):
        h = rotate_y_sphere_hit(f32(case["rot"][0]), f32(case["rot"][1]), c, r, o, d, f32(1e-3), tmax)
    else:
        h = sphere_hit(c, r, o, d, f32(1e-3), tmax)
    return _finish(case, h, d, key, depth)


def _finish(case, h, d, key, depth):
    if h is None:
        return {"hit": -1, "t": f32(0), "alive": False, "att": v3(1, 1, 1), "o": ZERO, "d": ZERO, "rad": sky_gradient(d)}
    mat = dict(case["mat"])
    if case.get("tex_eval"):
        mat["tex"] = case["tex_eval"](h)
    rng = Rng(key[0], key[1], depth)
    alive, att, so, sd, em = scatter(mat, d, h, rng)
    return {"hit": 0, "t": h["t"], "alive": alive, "att": att, "o": so, "d": sd, "rad": em}
