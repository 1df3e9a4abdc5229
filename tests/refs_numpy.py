"""Independent numpy readings of the reference's Rust text, written from /root/reference/src (not from the
oracle's C++): second opinions that pin the oracle — and, through it, the HIP path — where the reference ships no
rendered output (the lights / MIS branch, glass and metal pdf/eval).

Everything here is a closed form or a deterministic quadrature; nothing is sampled.
"""
from __future__ import annotations

import math

import numpy as np


# ---- bsdf/sampling.rs:38-55 ------------------------------------------------------------------------------
def ggx_D(h, roughness):
    cos_theta = max(h[2], 0.001)
    alpha2 = max(roughness * roughness, 0.001)
    denom = (alpha2 - 1.0) * (cos_theta * cos_theta) + 1.0
    return alpha2 / (math.pi * denom * denom)


def ggx_G1(w, roughness):
    alpha2 = max(roughness * roughness, 0.001)
    c = abs(w[2])
    return 2.0 * c / (c + math.sqrt(c * c * (1.0 - alpha2) + alpha2))


def dielectric_fresnel(w, h, eta_i, eta_o):   # glass.rs:51-62 (the same text as bsdf/mod.rs:77-88)
    c = abs(float(np.dot(w, h)))
    g_squared = (eta_o / eta_i) ** 2 - 1.0 + c * c
    if g_squared < 0.0:
        return 1.0
    g = math.sqrt(g_squared)
    gmc, gpc = g - c, g + c
    x = (c * gpc - 1.0) / (c * gmc + 1.0)
    return 0.5 * (gmc * gmc) / (gpc * gpc) * (1.0 + x * x)


def glass_pdf_eval(roughness, ior, v, l, front=True):
    """GlassBSDF::pdf glass.rs:92-123 and ::eval :125-163 for local vectors (normal +z). eval ignores base_color (Q4)."""
    v, l = np.asarray(v, float), np.asarray(l, float)
    reflect = l[2] * v[2] > 0.0
    eta_i, eta_o = (1.0, ior) if front else (ior, 1.0)
    if reflect:
        s = l + v
        h = s / np.linalg.norm(s) * math.copysign(1.0, v[2])
    else:
        s = l * eta_o + v * eta_i
        h = -(s / np.linalg.norm(s))
    d = ggx_D(h, roughness)
    g1v = ggx_G1(v, roughness)
    pdf_h = g1v * abs(float(v @ h)) * d / abs(v[2])
    f = dielectric_fresnel(v, h, eta_i, eta_o)
    v_dot_h, l_dot_h = float(v @ h), float(l @ h)
    if reflect:
        jacobian = f * 1.0 / (4.0 * abs(l_dot_h))
    else:
        jacobian = (1.0 - f) * (eta_o * eta_o * abs(l_dot_h)) / (eta_i * v_dot_h + eta_o * l_dot_h) ** 2
    g = g1v * ggx_G1(l, roughness)
    if reflect:
        factor = f * g * d / (4.0 * abs(l[2]) * abs(v[2]))
    else:
        term1 = abs((l_dot_h * v_dot_h) / (l[2] * v[2]))
        term2 = (eta_o * eta_o) / (eta_i * v_dot_h + eta_o * l_dot_h) ** 2
        factor = term1 * term2 * (1.0 - f) * g * d
    return pdf_h * jacobian, np.full(3, factor) * abs(l[2])


def metal_pdf_eval(base, roughness, v, l):
    """MetalBRDF::pdf metal.rs:56-67 and ::eval :69-80 (+ schlick_fresnel :108-110) for local vectors."""
    v, l, base = np.asarray(v, float), np.asarray(l, float), np.asarray(base, float)
    s = v + l
    h = s / np.linalg.norm(s)
    pdf_h = ggx_G1(v, roughness) * abs(float(v @ h)) * ggx_D(h, roughness) / abs(v[2])
    pdf = pdf_h * (1.0 / (4.0 * abs(float(l @ h))))
    g = ggx_G1(v, roughness) * ggx_G1(l, roughness)
    f = base + (1.0 - base) * (1.0 - float(l @ h)) ** 5
    return pdf, abs(l[2]) * (f * g * ggx_D(h, roughness) / (4.0 * abs(l[2]) * abs(v[2])))


# ---- camera.rs:51-77, 133-168 -----------------------------------------------------------------------------
def camera_frame(width, aspect, vfov_deg, look_from, look_at, vup, focal_length):
    height = int(width / aspect)
    look_from, look_at, vup = (np.asarray(a, float) for a in (look_from, look_at, vup))
    h = math.tan(math.radians(vfov_deg) / 2.0)
    viewport_height = 2.0 * h * focal_length
    viewport_width = viewport_height * (width / height)
    forward = look_from - look_at
    forward /= np.linalg.norm(forward)
    right = np.cross(vup, forward)
    right /= np.linalg.norm(right)
    up = np.cross(forward, right)
    viewport_u, viewport_v = right * viewport_width, up * -viewport_height
    du, dv = viewport_u / width, viewport_v / height
    upperleft = look_from - forward * focal_length - viewport_u / 2.0 - viewport_v / 2.0
    return {"height": height, "center": look_from, "pixel00": upperleft + (du + dv) * 0.5, "du": du, "dv": dv}


def pixel_footprint(n_rings=3, n_angles=8, blur_strength=0.5):
    """Deterministic quadrature nodes of random_offsets() * blur_strength (camera.rs:133-138,154): a point uniform in
    the disk of radius blur_strength (radius = sqrt(u1), angle = 2 pi u2) -> equal-weight nodes at the mid-area radii."""
    pts = []
    for i in range(n_rings):
        r = math.sqrt((i + 0.5) / n_rings) * blur_strength
        for j in range(n_angles):
            a = 2.0 * math.pi * (j + 0.5 * (i % 2)) / n_angles
            pts.append((r * math.cos(a), r * math.sin(a)))
    return np.array(pts)


def floor_points(frame, rows, cols, offsets, floor_y=0.0):
    """World points where the camera rays of the given pixels (+ blur offsets: x offset to the ROW, y to the COLUMN,
    camera.rs:155-157) meet the plane y = floor_y. Returns (n_pixels, n_offsets, 3). No lens (defocus_angle = 0)."""
    r = np.asarray(rows, float)[:, None] + offsets[None, :, 0]
    c = np.asarray(cols, float)[:, None] + offsets[None, :, 1]
    loc = frame["pixel00"][None, None, :] + frame["dv"][None, None, :] * r[..., None] + frame["du"][None, None, :] * c[..., None]
    o = frame["center"]
    d = loc - o
    t = (floor_y - o[1]) / d[..., 1]
    return o + d * t[..., None]


# ---- expected radiance of a Lambert floor point (normal +y) under one emitter, max_depth = 2 --------------------
# trace() camera.rs:170-228 with max_depth = 2: bounce 0 hits the floor (no emission), draws ONE direction — with
# probability 1/2 from lights.sample, else from the cosine lobe — weights it with brdf / (0.5 bsdf_pdf + 0.5 light_pdf);
# bounce 1 adds throughput * emitted if that direction reaches the emitter. Nothing after that is traced, so quirk Q5
# (a light-sampled path continuing through the emitter) cannot act.
def quad_light_floor_radiance(points, albedo, emission, q, u, v, n_quad=48):
    """Unbiased case: Quad::sample (uniform on the quad, quad.rs:80-86) and Quad::pdf (dist^2 / (cos * area), :88-98)
    agree, so E[estimate] = integral of f cos Le V dw = (albedo/pi) Le * integral over the quad of cos_x cos_l / r^2 dA."""
    q, u, v = (np.asarray(a, float) for a in (q, u, v))
    s = (np.arange(n_quad) + 0.5) / n_quad
    S, T = np.meshgrid(s, s, indexing="ij")
    pts = q + S[..., None] * u + T[..., None] * v                      # (n, n, 3)
    nl = np.cross(u, v)
    area = np.linalg.norm(nl)
    nl = nl / area
    dA = area / (n_quad * n_quad)
    P = np.asarray(points, float).reshape(-1, 3)
    form = np.empty(len(P))
    for a in range(0, len(P), 512):                                    # chunks keep the (points x nodes) temporaries small
        w = pts[None] - P[a:a + 512, None, None, :]
        r2 = np.sum(w * w, axis=-1)
        r = np.sqrt(r2)
        cos_x = np.maximum(w[..., 1] / r, 0.0)                         # floor normal +y
        cos_l = np.abs(np.sum(w * nl, axis=-1)) / r                    # the emitter radiates from both faces (material.rs:181-183)
        form[a:a + 512] = np.sum(cos_x * cos_l / r2, axis=(1, 2)) * dA
    return (np.asarray(albedo, float)[None, :] / math.pi) * np.asarray(emission, float)[None, :] * form[:, None]


def sphere_light_floor_radiance(points, albedo, emission, center, radius, n_alpha=160, n_phi=96):
    """Sphere::sample draws a point UNIFORM ON THE WHOLE SURFACE (sphere.rs:110-122) while Sphere::pdf reports
    1 / (2 pi sqrt(1 - r^2/d^2)) (sphere.rs:124-135) — neither the cone's solid angle nor the density of the sampler.
    Returns (expected value of the reference's estimator, true integral) per point:
        E = int_cone f cos Le * q(w) / (0.5 cos/pi + 0.5 p_ref) dw,   q = 0.5 cos/pi + 0.5 p_true(w),
        p_true(w) = (t1^2 + t2^2) / (4 pi r^2 |cos_s|)   (both surface points a direction passes through)."""
    P = np.asarray(points, float).reshape(-1, 3)
    c = np.asarray(center, float)
    axis = c[None, :] - P
    d = np.linalg.norm(axis, axis=1)
    axis = axis / d[:, None]
    a_max = np.arcsin(radius / d)                                      # half-angle of the cone
    # orthonormal frame around the axis
    helper = np.where(np.abs(axis[:, [0]]) > 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    e1 = np.cross(axis, helper); e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = np.cross(axis, e1)
    # Gauss-Legendre in alpha (the integrand has a 1/sqrt singularity of p_true at the rim: substitute alpha = a_max sin(s))
    xs, ws = np.polynomial.legendre.leggauss(n_alpha)
    s = (xs + 1.0) * (math.pi / 4.0); ws = ws * (math.pi / 4.0)       # s in (0, pi/2)
    phi = (np.arange(n_phi) + 0.5) * (2.0 * math.pi / n_phi)
    est = np.zeros((len(P), 3)); true = np.zeros((len(P), 3))
    alb, Le = np.asarray(albedo, float), np.asarray(emission, float)
    for i in range(len(P)):
        alpha = a_max[i] * np.sin(s)                                   # (n_alpha,)
        jac = a_max[i] * np.cos(s)                                     # d alpha / d s
        sa, ca = np.sin(alpha), np.cos(alpha)
        # direction w(alpha, phi)
        w = ca[:, None, None] * axis[i][None, None, :] + sa[:, None, None] * (np.cos(phi)[None, :, None] * e1[i][None, None, :] + np.sin(phi)[None, :, None] * e2[i][None, None, :])
        cos_x = np.maximum(w[..., 1], 0.0)                             # floor normal +y
        disc = np.maximum(radius * radius - (d[i] * sa) ** 2, 0.0)
        root = np.sqrt(disc)
        t1, t2 = d[i] * ca - root, d[i] * ca + root
        cos_s = root / radius                                          # |cos| between w and the surface normal at either point
        p_true = (t1 * t1 + t2 * t2) / (4.0 * math.pi * radius * radius * np.maximum(cos_s, 1e-300))
        p_ref = 1.0 / (2.0 * math.pi * math.sqrt(1.0 - radius * radius / (d[i] * d[i])))
        p_b = cos_x / math.pi
        q = 0.5 * p_b + 0.5 * p_true[:, None]
        ratio = q / (0.5 * p_b + 0.5 * p_ref)
        dw = (sa * jac * ws)[:, None] * (2.0 * math.pi / n_phi)        # sin(alpha) d alpha d phi
        k_est = np.sum(cos_x * ratio * dw)
        k_true = np.sum(cos_x * dw)
        est[i] = alb / math.pi * Le * k_est
        true[i] = alb / math.pi * Le * k_true
    return est, true


def _tri_rule(n):
    """Centroid rule on n^2 congruent sub-triangles as barycentric coordinates (m, 3); every node has weight area / n^2."""
    b = []
    for i in range(n):
        for j in range(n - i):
            b.append(((i + 1 / 3) / n, (j + 1 / 3) / n))
            if j < n - i - 1:
                b.append(((i + 2 / 3) / n, (j + 2 / 3) / n))
    b = np.array(b)
    return np.concatenate([1.0 - b.sum(axis=1, keepdims=True), b], axis=1)


def _ccw(poly):
    p = np.asarray(poly, float)
    x, y = p[:, 0], p[:, 1]
    return p if np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y) > 0 else p[::-1]


def _clip_convex(subject, clip):
    """Sutherland-Hodgman: the part of the convex polygon `subject` inside the convex CCW polygon `clip` ((k, 2), may be empty)."""
    out = [tuple(p) for p in subject]
    for k in range(len(clip)):
        a, b = clip[k], clip[(k + 1) % len(clip)]
        if not out:
            break
        side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        src, out = out, []
        for m in range(len(src)):
            p, q = src[m], src[(m + 1) % len(src)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return np.array(out).reshape(-1, 2)


def _poly_nodes(poly, rule):
    """Quadrature nodes (m, 2) and weights (m,) of a convex polygon: fan of triangles, `rule` on each."""
    nodes, weights = [], []
    for k in range(1, len(poly) - 1):
        tri = np.stack([poly[0], poly[k], poly[k + 1]])
        area = 0.5 * abs((tri[1, 0] - tri[0, 0]) * (tri[2, 1] - tri[0, 1]) - (tri[1, 1] - tri[0, 1]) * (tri[2, 0] - tri[0, 0]))
        if area > 0:
            nodes.append(rule @ tri)
            weights.append(np.full(len(rule), area / len(rule)))
    if not nodes:
        return np.zeros((0, 2)), np.zeros(0)
    return np.concatenate(nodes), np.concatenate(weights)


def coplanar_lights_floor_radiance(points, albedo, lights, eps=1e-3, n_big=28, n_strip=10):
    """E[reference estimator] at points of a Lambert floor (y = 0, normal +y) under a lights list of flat emitters lying
    in ONE horizontal plane y = h, max_depth = 2 (trace(), camera.rs:170-228). `lights`: dicts {"kind": "quad"|"tri",
    "verts": the quad's 4 corners in order | the triangle's 3 vertices, "emission": rgb}.

    What the reference does at the floor hit x0:
      * with probability 1/2 HittableList::sample picks ONE light uniformly (list.rs:78-84); Quad::sample is uniform on
        the quad (quad.rs:80-86); Triangle::sample draws u, v uniform in [0,1]^2, w = 1 - u - v (mesh.rs:122-129): the
        point is uniform over the PARALLELOGRAM of the two edges, half of it (the "outer" half O) outside the triangle;
      * the claimed pdf is 1/2 cos/pi + 1/2 MEAN over the lights of their pdf (list.rs:86-96), each dist^2 / (cos_l area)
        where the ray from x0 meets that light (quad.rs:88-98, mesh.rs:131-141 — the triangle claims its OWN area, twice
        the sampler's density);
      * the next segment starts at x0 + EPS n, not at x0 (camera.rs:217-222), with the SAME direction: it meets the plane
        at  x0_h + (1 - EPS/h)(hp - x0_h)  when the direction from x0 aims at hp. So the emitter is reached iff hp lies in
        the emitter scaled by 1/(1 - EPS/h) about x0_h, a polygon E' that differs from the emitter E by strips EPS/h wide.
        Where E' sticks out of E the claimed light pdf is 0 and the weight is brdf / (pdf_bsdf / 2) = 2 albedo: rare for
        cosine-sampled directions, but the triangle's sampler puts half its points in O, right across the long edge — a
        +2.4 % share at the test's geometry, which is why it has to be modelled (the wide-spread emitters of the other
        tests lose and gain ~1e-4 there).
    With g = cos_x cos_l / r^2 and, on a cell c of the plane, claimed density C_c and sampler density S_c (per area):
        E = albedo/pi * Le_i * [ integral over E'_i of g  +  sum over cells of integral over (E'_i ^ cell) of g (ratio_c - 1) ],
        ratio_c = (p_b/2 + S_c r^2/cos_l / (2n)) / (p_b/2 + C_c r^2/cos_l / (2n)),   p_b = cos_x / pi,  n = number of lights.
    Cells: a quad (C = S = 1/A: ratio 1, nothing to add), a triangle T (C = 1/A_t, S = 1/(2 A_t)), its outer half O
    (C = 0, S = 1/(2 A_t)). Returns (expected, the true integral = the first term)."""
    P = np.asarray(points, float).reshape(-1, 3)
    alb = np.asarray(albedo, float)
    h = float(np.asarray(lights[0]["verts"], float)[0][1])
    n_l = len(lights)
    big, strip = _tri_rule(n_big), _tri_rule(n_strip)
    emit, cells = [], []
    for L in lights:
        V = np.asarray(L["verts"], float)
        assert np.all(V[:, 1] == h), "all emitters in one horizontal plane"
        V2 = V[:, [0, 2]]
        if L["kind"] == "quad":
            emit.append(_ccw(V2))
        else:
            A = 0.5 * abs((V2[1, 0] - V2[0, 0]) * (V2[2, 1] - V2[0, 1]) - (V2[1, 1] - V2[0, 1]) * (V2[2, 0] - V2[0, 0]))
            emit.append(_ccw(V2))
            cells.append((_ccw(V2), 1.0 / A, 0.5 / A, big))
            cells.append((_ccw(np.stack([V2[1], V2[1] + V2[2] - V2[0], V2[2]])), 0.0, 0.5 / A, strip))

    def kernel(x0, nodes):
        d = np.stack([nodes[:, 0] - x0[0], np.full(len(nodes), h - x0[1]), nodes[:, 1] - x0[2]], axis=1)
        r2 = np.sum(d * d, axis=1)
        cos = np.abs(d[:, 1]) / np.sqrt(r2)                          # floor normal and emitter normals are both +-y
        return cos * cos / r2, cos, r2

    est = np.zeros((len(P), 3)); true = np.zeros((len(P), 3))
    s = 1.0 - eps / h
    for k, x0 in enumerate(P):
        c2 = np.array([x0[0], x0[2]])
        for L, E in zip(lights, emit):
            Ev = c2 + (E - c2) / s                                    # directions whose offset segment reaches the emitter
            nodes, w = _poly_nodes(Ev, big)
            g, _, _ = kernel(x0, nodes)
            first = float(np.sum(g * w))
            extra = 0.0
            for cell, C, S, rule in cells:
                piece = _clip_convex(Ev, cell)
                if len(piece) < 3:
                    continue
                nodes, w = _poly_nodes(piece, rule)
                g, cos, r2 = kernel(x0, nodes)
                p_b = cos / math.pi
                solid = r2 / cos / (2.0 * n_l)
                extra += float(np.sum(g * ((0.5 * p_b + S * solid) / (0.5 * p_b + C * solid) - 1.0) * w))
            le = np.asarray(L["emission"], float)
            true[k] += alb / math.pi * le * first
            est[k] += alb / math.pi * le * (first + extra)
    return est, true


# ---- Cuboid::{sample,pdf} (cuboid.rs:78-84 -> list.rs:78-96 over the six sides) and Instance::{sample,pdf} (instance.rs:64-75) ----
def box_light_floor_radiance(points, albedo, emission, lo, hi, n_quad=40):
    """A cuboid emitter [lo, hi] over a Lambert floor (normal +y), max_depth = 2. Cuboid::sample picks one of the SIX sides
    uniformly and a uniform point on it; Cuboid::pdf is the MEAN over the six sides of dist^2 / (|cos| area) for every side
    the ray from the floor point meets — entry AND exit side. A direction through the box is therefore proposed with density
    (1/6) sum over the sides it meets of r^2 / (|cos| A), which is exactly the claimed pdf: the one-sample MIS estimator is
    unbiased, whichever side a sampled point lay on (the next segment sees the entry side either way, and the emitter
    radiates from both faces, material.rs:181-183). So E = (albedo/pi) Le * integral over the box's solid angle of cos_x dw
    = sum over the sides FACING the point of the form factor (a convex body: each direction is counted once)."""
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    P = np.asarray(points, float).reshape(-1, 3)
    s = (np.arange(n_quad) + 0.5) / n_quad
    S, T = np.meshgrid(s, s, indexing="ij")
    form = np.zeros(len(P))
    for axis in range(3):
        a, b = (axis + 1) % 3, (axis + 2) % 3
        for side, coord in ((-1.0, lo[axis]), (1.0, hi[axis])):
            pts = np.zeros(S.shape + (3,))
            pts[..., axis] = coord
            pts[..., a] = lo[a] + S * (hi[a] - lo[a])
            pts[..., b] = lo[b] + T * (hi[b] - lo[b])
            n_out = np.zeros(3); n_out[axis] = side
            dA = (hi[a] - lo[a]) * (hi[b] - lo[b]) / (n_quad * n_quad)
            for k in range(0, len(P), 512):
                w = pts[None] - P[k:k + 512, None, None, :]
                r2 = np.sum(w * w, axis=-1)
                r = np.sqrt(r2)
                cos_x = np.maximum(w[..., 1] / r, 0.0)
                cos_l = np.maximum(-(w @ n_out) / r, 0.0)                 # only the sides whose OUTWARD normal faces the point
                form[k:k + 512] += np.sum(cos_x * cos_l / r2, axis=(1, 2)) * dA
    return (np.asarray(albedo, float)[None, :] / math.pi) * np.asarray(emission, float)[None, :] * form[:, None]


def rigid_quad(q, u, v, axis, angle, translation):
    """Instance::new(object, axis, angle, translation) (instance.rs:20-30): rotate about `axis` by `angle` (right-handed,
    Rodrigues' formula — not the quaternion route of glam that the restatements take), then translate. Returns the world-space
    (q, u, v) of a quad wrapped by that instance."""
    k = np.asarray(axis, float)
    k = k / np.linalg.norm(k)
    K = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    Rm = np.eye(3) + math.sin(angle) * K + (1.0 - math.cos(angle)) * (K @ K)
    q, u, v = (np.asarray(a, float) for a in (q, u, v))
    return Rm @ q + np.asarray(translation, float), Rm @ u, Rm @ v
