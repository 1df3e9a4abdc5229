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
