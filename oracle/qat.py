"""CPU restatement of the QAT getters -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What it restates (numpy, every operation a single fp32 op in the order written):
  * torch.ao.quantization.FakeQuantize(dtype=torch.qint8) as the reference instantiates it
    (scene/gaussian_model.py:109-118): MovingAverageMinMaxObserver.forward (averaging constant 0.01),
    UniformQuantizationObserverBase._calculate_qparams (per_tensor_affine, [-128, 127], eps = fp32 epsilon) and
    fake_quantize_per_tensor_affine with its straight-through mask (ATen fake_quantize_core);
  * the getters of GaussianModel (scene/gaussian_model.py:54-77, 213-267) and FakeQuantizationHalf (:1405-1414);
  * the [visible] gathers of GaussianModel.render (:851-862).
Pinning: the algorithm lives in a third-party dependency (torch 2.10, importable in this image, CPU), so
tests/test_oracle_qat.py runs the real torch modules / ops next to this file on the same inputs, several observer
steps deep (state evolution), forward and backward. Activations that torch evaluates with vectorised
transcendental code (sigmoid, exp) agree to 1-2 ulp, not bitwise; fake-quantised outputs therefore may differ by one
quantisation step on a vanishing fraction of elements, which the tests bound explicitly.
"""
import numpy as np

F32 = np.float32
EPS = np.finfo(np.float32).eps
QMIN, QMAX = -128, 127
NORM_EPS = F32(1e-12)


class FqState:
    """min_val / max_val / scale / zero_point of one module (c3dgs_fq_state)."""

    def __init__(self):
        self.min_val, self.max_val, self.scale, self.zero_point = F32(np.inf), F32(-np.inf), F32(1.0), 0

    def copy(self):
        c = FqState()
        c.min_val, c.max_val, c.scale, c.zero_point = self.min_val, self.max_val, self.scale, self.zero_point
        return c

    def as_row(self):
        row = np.zeros(4, np.float32)
        row[0], row[1], row[2] = self.min_val, self.max_val, self.scale
        row[3:4].view(np.int32)[0] = self.zero_point
        return row


def observe(st, x, c=0.01):
    """MovingAverageMinMaxObserver.forward + calculate_qparams (FakeQuantize.forward, observer half)."""
    x = np.asarray(x, np.float32)
    if x.size == 0:
        return st
    lo, hi = F32(x.min()), F32(x.max())
    c = F32(c)
    if st.min_val == F32(np.inf) and st.max_val == F32(-np.inf):
        st.min_val, st.max_val = lo, hi
    else:
        st.min_val = F32(st.min_val + F32(c * F32(lo - st.min_val)))
        st.max_val = F32(st.max_val + F32(c * F32(hi - st.max_val)))
    if st.min_val == F32(np.inf) and st.max_val == F32(-np.inf):
        st.scale, st.zero_point = F32(1.0), 0
        return st
    min_neg, max_pos = min(st.min_val, F32(0)), max(st.max_val, F32(0))
    st.scale = max(F32(F32(max_pos - min_neg) / F32(QMAX - QMIN)), EPS)
    zp = QMIN - int(np.rint(F32(min_neg / st.scale)))
    st.zero_point = int(min(QMAX, max(QMIN, zp)))
    return st


def fake_quant(st, x, enabled=True):
    """-> (y, mask): fake_quantize_per_tensor_affine and its gradient mask."""
    x = np.asarray(x, np.float32)
    if not enabled:
        return x.copy(), np.ones(x.shape, bool)
    inv = F32(F32(1.0) / st.scale)
    q = np.rint(x * inv).astype(np.float32) + F32(st.zero_point)
    mask = (q >= QMIN) & (q <= QMAX)
    y = (np.clip(q, QMIN, QMAX).astype(np.float32) - F32(st.zero_point)) * st.scale
    return y.astype(np.float32), mask


def sigmoid(x):
    x = np.asarray(x, np.float32)
    return (F32(1.0) / (F32(1.0) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def normalize_rows(x):
    """torch.nn.functional.normalize(x, dim=1): x / max(||x||_2, 1e-12)."""
    x = np.asarray(x, np.float32)
    acc = np.zeros(x.shape[0], np.float32)
    for k in range(x.shape[1]):
        acc = acc + x[:, k] * x[:, k]
    n = np.sqrt(acc, dtype=np.float32)
    d = np.maximum(n, NORM_EPS)
    return (x / d[:, None]).astype(np.float32), n, d


def half_round(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


SLOTS = ("opacity", "scaling", "scaling_factor", "rotation", "features_dc", "features_rest")


class Getters:
    """The getters the indexed QAT render reads, with the six module states (one observer step per `forward`)."""

    def __init__(self, quantization=True):
        self.st = {k: FqState() for k in SLOTS}
        self.observer = {k: True for k in SLOTS}
        self.fq = {k: True for k in SLOTS}
        self.half_xyz = True
        if not quantization:                      # gaussian_model.py:120-134 (opacity_qa stays enabled there)
            for k in ("features_dc", "features_rest", "scaling", "scaling_factor", "rotation"):
                self.observer[k] = self.fq[k] = False
            self.half_xyz = False

    def _module(self, name, x):
        if self.observer[name]:
            observe(self.st[name], x)
        return fake_quant(self.st[name], x, self.fq[name])

    def forward(self, xyz, opacity, scaling_factor, scaling, rotation, fdc, frest):
        """-> dict of full-size getter outputs plus the intermediates the backward needs."""
        o = {}
        o["xyz"] = half_round(xyz) if self.half_xyz else np.asarray(xyz, np.float32).copy()
        sg = sigmoid(opacity)
        o["sig"] = sg
        o["opacity"], o["m_opacity"] = self._module("opacity", sg)
        u = np.maximum(np.asarray(scaling, np.float32), F32(0))
        v, n, d = normalize_rows(u)
        o["u"], o["v"], o["un"], o["ud"] = u, v, n, d
        o["scales_n"], o["m_scaling"] = self._module("scaling", v)
        sfq, o["m_sf"] = self._module("scaling_factor", scaling_factor)
        o["scale_factors"] = np.exp(sfq, dtype=np.float32)
        w, o["m_rot"] = self._module("rotation", rotation)
        o["w"] = w
        o["rotations"], o["wn"], o["wd"] = normalize_rows(w)
        dc, o["m_dc"] = self._module("features_dc", fdc)
        if frest is not None and np.asarray(frest).shape[1] > 0:
            rest, o["m_rest"] = self._module("features_rest", frest)
            o["shs"] = np.concatenate([dc, rest], axis=1)
        else:
            o["shs"], o["m_rest"] = dc, None
        return o

    @staticmethod
    def backward(o, scaling, g_opacity=None, g_sf=None, g_scales_n=None, g_rot=None, g_shs=None):
        """Gradients w.r.t. the raw tensors (float64 accumulation of the small dot products)."""
        r = {}
        if g_opacity is not None:
            r["opacity"] = (g_opacity * o["m_opacity"] * (1.0 - o["sig"].astype(np.float64)) * o["sig"]).astype(np.float32)
        if g_sf is not None:
            r["scaling_factor"] = (g_sf * o["m_sf"] * o["scale_factors"].astype(np.float64)).astype(np.float32)
        if g_scales_n is not None:
            dv = (g_scales_n * o["m_scaling"]).astype(np.float64)
            u, n, d = o["u"].astype(np.float64), o["un"].astype(np.float64), o["ud"].astype(np.float64)
            dot = (u * dv).sum(1)
            k2 = np.where((n >= 1e-12) & (n > 0), dot / np.where(n > 0, n * d * d, 1.0), 0.0)
            du = dv / d[:, None] - u * k2[:, None]
            r["scaling"] = (du * (np.asarray(scaling) > 0)).astype(np.float32)
        if g_rot is not None:
            w, n, d = o["w"].astype(np.float64), o["wn"].astype(np.float64), o["wd"].astype(np.float64)
            g = g_rot.astype(np.float64)
            dot = (w * g).sum(1)
            k2 = np.where((n >= 1e-12) & (n > 0), dot / np.where(n > 0, n * d * d, 1.0), 0.0)
            r["rotation"] = ((g / d[:, None] - w * k2[:, None]) * o["m_rot"]).astype(np.float32)
        if g_shs is not None:
            r["features_dc"] = (g_shs[:, :1] * o["m_dc"]).astype(np.float32)
            if o["m_rest"] is not None:
                r["features_rest"] = (g_shs[:, 1:] * o["m_rest"]).astype(np.float32)
        return r


def visible_rows(xyz_q, viewmatrix):
    """rasterizer markVisible on get_xyz: view-space z > 0.01 with the row-vector matrix convention of the reference
    (rasterizer_impl.cu:54-66 + auxiliary.h:139-166); same fp32 op order as the oracle's in_frustum."""
    m = np.asarray(viewmatrix, np.float32).reshape(-1)
    x, y, z = xyz_q[:, 0], xyz_q[:, 1], xyz_q[:, 2]
    pz = (m[2] * x + m[6] * y).astype(np.float32)
    pz = (pz + m[10] * z).astype(np.float32)
    pz = (pz + m[14]).astype(np.float32)
    return ~(pz <= F32(0.01))


def quantize_codes(st, x, device_rounding=True):
    """torch.quantize_per_tensor(x, scale, zero_point, qint8).int_repr() (save_npz, scene/gaussian_model.py:525-617).
    torch's device kernel rounds nearbyint(double(x) / double(scale)) (device_rounding=True; what the reference's GPU
    save path produces, verified on the MI355X box); its CPU path rounds fp32 x * (1/scale) (device_rounding=False,
    pinned bit-exact against CPU torch in tests/test_oracle_qat.py). The two differ only at rounding ties."""
    x = np.asarray(x, np.float32)
    if device_rounding:
        q = np.rint(x.astype(np.float64) / np.float64(st.scale)) + st.zero_point
    else:
        q = np.rint(x * F32(F32(1.0) / st.scale)).astype(np.float64) + st.zero_point
    return np.clip(q, QMIN, QMAX).astype(np.int8)
