"""A SECOND fp32 evaluation of the alpha blend (test infrastructure, numpy): the same decisions as the oracle's
blend_pixel_fwd (forward.cu:330-360) but with `power` rounded the way render.hip rounds it -- conic pre-scaled by log2(e),
three products and three fused multiply-adds, 2^x -- so that CPU tests can produce genuinely "flipped" pixels (two correct
fp32 evaluations disagreeing on a threshold decision) for tests/fullsize.py:prove_flips without a GPU.

FMAs are emulated as float64 arithmetic rounded once to fp32 (the product of two fp32 values is exact in float64; the sum's
double rounding is irrelevant here: ANY evaluation inside the error bound of prove_flips is a legitimate second opinion)."""
import numpy as np

_f = np.float32


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(_f)


def forward(st):
    """-> dict(out_color [3,H,W], final_T [N], n_contrib [N]) like tests/gpu_util.unpack gives for the HIP path."""
    W, H = st.W, st.H
    gx = (W + 15) // 16
    feat = (st.inputs["colors_precomp"] if st.inputs["colors_precomp"] is not None else st.rgb).reshape(st.P, 3)
    bg = st.inputs["bg"].astype(_f)
    out = np.zeros((3, H, W), _f)
    final_T = np.ones(W * H, _f)
    n_contrib = np.zeros(W * H, np.uint32)
    LOG2E = _f(1.4426950408889634)
    for t in range(st.ranges.shape[0]):
        ty, tx = divmod(t, gx)
        ys, xs = np.meshgrid(np.arange(ty * 16, min(ty * 16 + 16, H)), np.arange(tx * 16, min(tx * 16 + 16, W)), indexing="ij")
        ys, xs = ys.ravel(), xs.ravel()
        npx = ys.size
        T = np.ones(npx, _f)
        C = np.zeros((npx, 3), _f)
        last = np.zeros(npx, np.uint32)
        done = np.zeros(npx, bool)
        lst = st.point_list[st.ranges[t, 0]:st.ranges[t, 1]]
        for j, gid in enumerate(lst):
            if done.all():
                break
            mx, my = st.means2D[gid]
            a, b, c, op = st.conic_opacity[gid]
            ka, kb, kc = _f(a * _f(_f(-0.5) * LOG2E)), _f(b * _f(-LOG2E)), _f(c * _f(_f(-0.5) * LOG2E))
            dx, dy = (mx - xs.astype(_f)).astype(_f), (my - ys.astype(_f)).astype(_f)
            p2 = _fma(np.full(npx, kb, _f), (dx * dy).astype(_f), _fma(np.full(npx, kc, _f), (dy * dy).astype(_f), (ka * (dx * dx).astype(_f)).astype(_f)))
            G = np.exp2(np.minimum(p2, _f(60)).astype(np.float64)).astype(_f)
            alpha = np.minimum(_f(0.99), (op * G).astype(_f))
            hit = ~(p2 > 0) & ~(alpha < _f(1.0) / _f(255.0)) & ~done
            test_T = (T * (_f(1) - alpha)).astype(_f)
            stop = hit & (test_T < _f(0.0001))
            blend = hit & ~stop
            w = np.where(blend, (alpha * T).astype(_f), _f(0))
            C = (C + feat[gid][None, :] * w[:, None]).astype(_f)
            T = np.where(blend, test_T, T)
            last = np.where(blend, np.uint32(j + 1), last)
            done |= stop
        pix = ys * W + xs
        final_T[pix] = T
        n_contrib[pix] = last
        out[:, ys, xs] = (C + T[:, None] * bg[None, :]).T
    return dict(out_color=out, final_T=final_T, n_contrib=n_contrib)
