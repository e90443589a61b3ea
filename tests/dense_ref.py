"""Float64 torch-autograd re-derivation of the rasterizer's continuous math (TEST INFRASTRUCTURE).

Independent of oracle/c3dgs_oracle.c: written from the equations (EWA projection, SH evaluation,
front-to-back alpha blending), differentiated by torch.autograd in float64.  It borrows only the
DISCRETE structure from an oracle forward (which Gaussians are in which tile, in which order), so
it pins the oracle's analytic backward (K10-K12) and forward colours on tiny scenes.

Reference semantics mirrored on purpose:
  * alpha = min(0.99, o*G); skip power>0, alpha<1/255; stop before blending when T(1-alpha)<1e-4
  * colour = SH + 0.5, clamped at 0 only if clamp_color
  * quaternions are NOT normalised inside the op
  * the t-clamp of computeCov2D passes no gradient through the clamped coordinate
  * dL/dmeans2D is in NDC units (0.5*W, 0.5*H factors)
"""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def _sh_color(deg, sh, dirs):
    """sh [P,M,3], dirs [P,3] unit -> [P,3] (before +0.5)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
               + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def _rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    return R


def dense_render(st, leaves, dL_dout=None):
    """st: oracle RasterState (for the discrete structure + camera); leaves: dict of float64 tensors
    (requires_grad as wanted): means3D, means2D(zeros[P,3]), opacities[P], and one of shs|colors_precomp,
    one of (scales, rotations[, scale_factors])|cov3D_precomp.  Returns image [3,H,W] (float64)."""
    i = st.inputs
    dd = torch.float64
    view = torch.tensor(i["viewmatrix"], dtype=dd).reshape(4, 4)   # = W2C^T
    proj = torch.tensor(i["projmatrix"], dtype=dd).reshape(4, 4)
    campos = torch.tensor(i["campos"], dtype=dd)
    W, H = st.W, st.H
    tfx, tfy = float(i["tan_fovx"]), float(i["tan_fovy"])
    fx, fy = W / (2.0 * tfx), H / (2.0 * tfy)
    mean = leaves["means3D"]
    P = mean.shape[0]
    ones = torch.ones(P, 1, dtype=dd)
    hom = torch.cat([mean, ones], 1) @ proj                       # row-vector convention (matrix is transposed)
    p_w = 1.0 / (hom[:, 3] + 0.0000001)
    ndc = hom[:, :2] * p_w[:, None]
    t = (torch.cat([mean, ones], 1) @ view)[:, :3]

    if "cov3D_precomp" in leaves:
        c6 = leaves["cov3D_precomp"]
        Sigma = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4], c6[:, 5]],
                            -1).reshape(-1, 3, 3)
    else:
        sc, rt = leaves["scales"], leaves["rotations"]
        mod = float(i["scale_modifier"])
        if i["g_indices"] is not None:
            gi = torch.as_tensor(i["g_indices"])
            s = sc[gi] * (leaves["scale_factors"].reshape(-1, 1) * mod)
            Rm = _rot(rt[gi])
        else:
            s = sc * mod
            Rm = _rot(rt)
        Lm = Rm * s[:, None, :]
        Sigma = Lm @ Lm.transpose(1, 2)

    limx, limy = 1.3 * tfx, 1.3 * tfy
    txtz, tytz = t[:, 0] / t[:, 2], t[:, 1] / t[:, 2]
    cx = (txtz < -limx) | (txtz > limx)
    cy = (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * t[:, 2]).detach(), t[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * t[:, 2]).detach(), t[:, 1])
    tz = t[:, 2]
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], -1).reshape(-1, 2, 3)
    Rw = view[:3, :3].T                                           # W2C rotation
    A = J @ Rw
    cov = A @ Sigma @ A.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    con = torch.stack([c / det, -b / det, a / det], -1)

    m2d = leaves["means2D"]
    pixx = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5 + 0.5 * W * m2d[:, 0]
    pixy = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5 + 0.5 * H * m2d[:, 1]

    if "colors_precomp" in leaves:
        col = leaves["colors_precomp"]
    else:
        sh = leaves["shs"]
        if i["sh_indices"] is not None:
            sh = sh[torch.as_tensor(i["sh_indices"])]
        d = mean - campos
        d = d / d.norm(dim=1, keepdim=True)
        col = _sh_color(int(i["degree"]), sh, d) + 0.5
        if i["clamp_color"]:
            col = torch.clamp_min(col, 0.0)

    opac = leaves["opacities"].reshape(-1)
    bg = torch.tensor(i["bg"], dtype=dd)
    img = torch.zeros(3, H, W, dtype=dd)
    gx = (W + 15) // 16
    for tile in range(st.T):
        r0, r1 = int(st.ranges[tile, 0]), int(st.ranges[tile, 1])
        tx0, ty0 = (tile % gx) * 16, (tile // gx) * 16
        xs = torch.arange(tx0, min(tx0 + 16, W), dtype=dd)
        ys = torch.arange(ty0, min(ty0 + 16, H), dtype=dd)
        if len(xs) == 0 or len(ys) == 0:
            continue
        py, px = torch.meshgrid(ys, xs, indexing="ij")
        T = torch.ones_like(px)
        Cc = torch.zeros(3, *px.shape, dtype=dd)
        done = torch.zeros_like(px, dtype=torch.bool)
        for k in range(r0, r1):
            g = int(st.point_list[k])
            dx, dy = pixx[g] - px, pixy[g] - py
            power = -0.5 * (con[g, 0] * dx * dx + con[g, 2] * dy * dy) - con[g, 1] * dx * dy
            G = torch.exp(power)
            alpha = torch.clamp_max(opac[g] * G, 0.99)
            ok = (power <= 0) & (alpha >= 1.0 / 255.0) & ~done
            test_T = T * (1 - alpha)
            newly = ok & (test_T < 0.0001)
            done = done | newly
            contrib = ok & ~newly
            Cc = Cc + torch.where(contrib, alpha * T, torch.zeros_like(T))[None] * col[g][:, None, None]
            T = torch.where(contrib, test_T, T)
        y0, x0 = int(ys[0]), int(xs[0])
        img[:, y0:y0 + len(ys), x0:x0 + len(xs)] = Cc + T[None] * bg[:, None, None]
    return img
