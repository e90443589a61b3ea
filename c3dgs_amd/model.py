"""The QAT getters and render glue of the reference's GaussianModel, fused on the device.

Mirrors, for the render path only, scene/gaussian_model.py:
    setup_functions / activations                 :54-77
    the FakeQuantize(dtype=qint8) module set      :109-134
    get_scaling ... get_opacity                   :213-267
    GaussianModel.render                          :766-886
    FakeQuantizationHalf                          :1405-1414
Densification and ply IO of the reference class are outside the hot path and not mirrored.

The reference evaluates every getter with torch ops and seven torch.ao FakeQuantize modules: about a hundred small
launches and twenty host syncs per view (aminmax + float(scale) / int(zero_point) per module, one nonzero per
boolean-mask gather). Here the module state (min, max, scale, zero_point) lives in ONE device tensor that is never
read back, and an indexed QAT render is: visible flags + scan, ONE observe pass, ONE codebook launch, ONE compaction
launch, the rasterizer, and two backward launches (csrc/qat.hip). A single 4-byte device->host read (the visible count)
remains; it overlaps the observe / codebook kernels.
"""
import ctypes as C

import torch

from . import _lib
from . import rasterizer as _rz
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, GaussianRasterizerIndexed

SLOTS = ("opacity", "scaling", "scaling_factor", "rotation", "features_dc", "features_rest")
_SLOT = {k: i for i, k in enumerate(SLOTS)}
AVERAGING_CONSTANT = 0.01       # MovingAverageMinMaxObserver default


class ColorMode:                # scene/gaussian_model.py:49-51
    NOT_INDEXED = 0
    ALL_INDEXED = 1


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _need_gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"c3dgs_amd: {name} must be a GPU tensor (there is no CPU path)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    return t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def new_fq_state(device, n=1):
    """[n,4] float32 rows {min_val=+inf, max_val=-inf, scale=1, zero_point=0 (int32 bits)} -- c3dgs_fq_state."""
    s = torch.zeros(n, 4, dtype=torch.float32, device=device)
    s[:, 0] = float("inf")
    s[:, 1] = float("-inf")
    s[:, 2] = 1.0
    return s


_WS = {}


def _workspace(device):
    dev = torch.device(device)
    ws = _WS.get(dev)
    if ws is None:
        ws = torch.empty(_lib.lib().c3dgs_qat_workspace_bytes(), dtype=torch.uint8, device=dev)
        _WS[dev] = ws
    return ws


class _Observer:
    """`module.activation_post_process` look-alike: min_val / max_val views of the device state."""

    def __init__(self, row):
        self._row = row

    @property
    def min_val(self):
        return self._row[0]

    @property
    def max_val(self):
        return self._row[1]

    averaging_constant = AVERAGING_CONSTANT
    quant_min, quant_max = -128, 127


class _FakeQuantizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module):
        xc = _need_gpu(x, "FakeQuantize input")
        out = torch.empty_like(xc)
        row = module._row
        _lib.check(_lib.lib().c3dgs_fake_quantize(xc.numel(), xc.data_ptr(), row.data_ptr(), int(module.observer_enabled),
                                                  int(module.fake_quant_enabled), AVERAGING_CONSTANT, out.data_ptr(),
                                                  _workspace(xc.device).data_ptr(), _stream(xc.device)))
        ctx.enabled = int(module.fake_quant_enabled)
        ctx.save_for_backward(xc, row.clone())
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        x, row = ctx.saved_tensors
        gc = g.contiguous()
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().c3dgs_fake_quantize_backward(x.numel(), x.data_ptr(), row.data_ptr(), ctx.enabled, gc.data_ptr(),
                                                           dx.data_ptr(), _stream(x.device)))
        return dx, None


class FakeQuantize:
    """Device-resident stand-in for torch.ao.quantization.FakeQuantize(dtype=torch.qint8) as the reference builds it
    (scene/gaussian_model.py:109-118): MovingAverageMinMaxObserver, per-tensor affine, [-128, 127]. Same call and the
    same attribute names the reference touches; the state is one 16-byte row on the device."""

    quant_min, quant_max = -128, 127
    dtype = torch.qint8

    def __init__(self, dtype=torch.qint8, device="cuda", _row=None):
        if dtype != torch.qint8:
            raise RuntimeError("c3dgs_amd.model.FakeQuantize mirrors the reference's dtype=torch.qint8 modules only")
        self._row = _row if _row is not None else new_fq_state(device)[0]
        self.observer_enabled = True
        self.fake_quant_enabled = True
        self.activation_post_process = _Observer(self._row)

    def to(self, device):
        if torch.device(device) != self._row.device:
            self._row = self._row.to(device)
            self.activation_post_process = _Observer(self._row)
        return self

    @property
    def scale(self):
        return self._row[2:3]

    @property
    def zero_point(self):
        return self._row[3:4].view(torch.int32)

    def calculate_qparams(self):
        return self.scale, self.zero_point

    def enable_fake_quant(self, enabled=True):
        self.fake_quant_enabled = bool(enabled)
        return self

    def disable_fake_quant(self):
        return self.enable_fake_quant(False)

    def enable_observer(self, enabled=True):
        self.observer_enabled = bool(enabled)
        return self

    def disable_observer(self):
        return self.enable_observer(False)

    def __call__(self, x):
        if x.numel() == 0:
            return x
        return _FakeQuantizeFn.apply(x, self)

    forward = __call__


class FakeQuantizationHalf(torch.autograd.Function):
    """scene/gaussian_model.py:1405-1414: round through fp16, identity gradient."""

    @staticmethod
    def forward(_, x):
        return x.half().float()

    @staticmethod
    def backward(_, grad_output):
        return grad_output


# ----------------------------------------------------------------------------- fused getters
def _qat_params(model, state, *, xyz=None, opacity=None, scaling_factor=None, scaling=None, rotation=None, fdc=None,
                frest=None):
    q = _lib.QatParams()
    q.P = int((xyz if xyz is not None else opacity if opacity is not None else scaling_factor).shape[0]) \
        if (xyz is not None or opacity is not None or scaling_factor is not None) else 0
    q.GS = int((scaling if scaling is not None else rotation).shape[0]) if (scaling is not None or rotation is not None) else 0
    q.SHS = int(fdc.shape[0]) if fdc is not None else 0
    q.M = 1 + (int(frest.shape[1]) if frest is not None else 0)
    q.xyz, q.opacity, q.scaling_factor = _ptr(xyz), _ptr(opacity), _ptr(scaling_factor)
    q.scaling, q.rotation = _ptr(scaling), _ptr(rotation)
    q.features_dc = _ptr(fdc)
    q.features_rest = _ptr(frest) if (frest is not None and frest.numel() > 0) else None
    q.state = state.data_ptr()
    for i, k in enumerate(SLOTS):
        m = model._modules_qa[k]
        q.observer_enabled[i] = int(m.observer_enabled)
        q.fake_quant_enabled[i] = int(m.fake_quant_enabled)
    q.half_xyz = int(model.quantization)
    q.averaging_constant = AVERAGING_CONSTANT
    return q


class _QatGetters(torch.autograd.Function):
    """All getters of one render in one autograd node. Inputs that are None are skipped; `vis` is None (plain getters,
    every row) or (visible u8[P], rank i32[P], V)."""

    @staticmethod
    def forward(ctx, model, vis, sh_indices, g_indices, xyz, screenspace, opacity, scaling_factor, scaling, rotation, fdc,
                frest):
        lib = _lib.lib()
        raw = dict(xyz=xyz, opacity=opacity, scaling_factor=scaling_factor, scaling=scaling, rotation=rotation, fdc=fdc,
                   frest=frest)
        raw = {k: (None if v is None else _need_gpu(v, k)) for k, v in raw.items()}
        anyt = next(v for v in raw.values() if v is not None)
        dev = anyt.device
        s = _stream(dev)
        q = _qat_params(model, model._fq_state, **raw)
        _lib.check(lib.c3dgs_qat_observe(C.byref(q), _workspace(dev).data_ptr(), s))
        state = model._fq_state.clone()                       # what the backward must see (the next view moves it)
        f32 = dict(dtype=torch.float32, device=dev)
        scales_n = torch.empty(q.GS, 3, **f32) if raw["scaling"] is not None else None
        rotations = torch.empty(q.GS, 4, **f32) if raw["rotation"] is not None else None
        shs = torch.empty(q.SHS, q.M, 3, **f32) if raw["fdc"] is not None else None
        if scales_n is not None or rotations is not None or shs is not None:
            _lib.check(lib.c3dgs_qat_codebooks(C.byref(q), _ptr(scales_n), _ptr(rotations), _ptr(shs), s))
        visible = rank = None
        V = q.P
        if vis is not None:
            visible, rank, V = vis
            V = int(V() if callable(V) else V)                # the one host read, deferred until here
        means3D = torch.empty(V, 3, **f32) if raw["xyz"] is not None else None
        # means2D only exists to carry a gradient back to `screenspace`: the rasterizer never reads its values
        means2D = torch.empty(V, 3, **f32) if screenspace is not None else None
        opac = torch.empty(V, 1, **f32) if raw["opacity"] is not None else None
        sfac = torch.empty(V, 1, **f32) if raw["scaling_factor"] is not None else None
        sh_out = torch.empty(V, dtype=torch.int64, device=dev) if (sh_indices is not None and vis is not None) else None
        g_out = torch.empty(V, dtype=torch.int64, device=dev) if (g_indices is not None and vis is not None) else None
        if q.P > 0 and (means3D is not None or opac is not None or sfac is not None or sh_out is not None):
            _lib.check(lib.c3dgs_qat_points(C.byref(q), _ptr(visible), _ptr(rank),
                                            _ptr(sh_indices) if sh_out is not None else None,
                                            _ptr(g_indices) if g_out is not None else None,
                                            _ptr(means3D), _ptr(opac), _ptr(sfac), _ptr(sh_out), _ptr(g_out), s))
        if vis is None:
            sh_out, g_out = sh_indices, g_indices
        ctx.model, ctx.P, ctx.has_screen = model, q.P, screenspace is not None
        ctx.flags = ([int(v) for v in q.observer_enabled], [int(v) for v in q.fake_quant_enabled], int(q.half_xyz))
        ctx.save_for_backward(state, visible, rank, *[raw[k] for k in ("xyz", "opacity", "scaling_factor", "scaling",
                                                                       "rotation", "fdc", "frest")])
        outs = (means3D, means2D, opac, sfac, scales_n, rotations, shs, sh_out, g_out)
        ctx.mark_non_differentiable(*[t for t in (sh_out, g_out) if t is not None and vis is not None])
        # without this, autograd hands backward zero-filled int64[V] "gradients" for the two index outputs on every call
        # (2 x 24 MB of fills at V = 3M); a differentiable output nobody used arrives as None and is filled below
        ctx.set_materialize_grads(False)
        ctx.V = V
        return outs

    @staticmethod
    def backward(ctx, g_m3, g_m2, g_op, g_sf, g_scales, g_rot, g_shs, _a, _b):
        lib = _lib.lib()
        state, visible, rank, xyz, opacity, sfac, scaling, rotation, fdc, frest = ctx.saved_tensors
        raw = dict(xyz=xyz, opacity=opacity, scaling_factor=sfac, scaling=scaling, rotation=rotation, fdc=fdc, frest=frest)
        q = _qat_params(ctx.model, state, **raw)
        for i in range(6):                                    # the flags as they were at forward time
            q.observer_enabled[i], q.fake_quant_enabled[i] = ctx.flags[0][i], ctx.flags[1][i]
        q.half_xyz = ctx.flags[2]
        anyt = next(v for v in raw.values() if v is not None)
        dev = anyt.device
        s = _stream(dev)
        f32 = dict(dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad                           # (model, vis, sh_idx, g_idx, xyz, screen, op, sf, scal, rot, dc, rest)
        c = lambda t: None if t is None else t.contiguous()
        g_m3, g_m2, g_op, g_sf, g_scales, g_rot, g_shs = map(c, (g_m3, g_m2, g_op, g_sf, g_scales, g_rot, g_shs))
        P = ctx.P
        if g_m3 is None and xyz is not None and need[4]:
            g_m3 = torch.zeros(ctx.V, 3, **f32)
        if g_m2 is None and ctx.has_screen and need[5]:
            g_m2 = torch.zeros(ctx.V, 3, **f32)
        if g_op is None and opacity is not None and need[6]:
            g_op = torch.zeros(ctx.V, 1, **f32)
        if g_sf is None and sfac is not None and need[7]:
            g_sf = torch.zeros(ctx.V, 1, **f32)
        d_xyz = torch.empty(P, 3, **f32) if (xyz is not None and need[4]) else None
        d_screen = torch.empty(P, 3, **f32) if (ctx.has_screen and need[5]) else None
        d_op = torch.empty(P, 1, **f32) if (opacity is not None and need[6]) else None
        d_sf = torch.empty(P, 1, **f32) if (sfac is not None and need[7]) else None
        if P > 0 and any(t is not None for t in (d_xyz, d_screen, d_op, d_sf)):
            _lib.check(lib.c3dgs_qat_points_backward(C.byref(q), _ptr(visible), _ptr(rank), _ptr(g_m3), _ptr(g_m2), _ptr(g_op),
                                                     _ptr(g_sf), _ptr(d_xyz), _ptr(d_screen), _ptr(d_op), _ptr(d_sf), s))
        d_scaling = torch.empty_like(scaling) if (scaling is not None and need[8]) else None
        d_rot = torch.empty_like(rotation) if (rotation is not None and need[9]) else None
        want_sh = fdc is not None and (need[10] or need[11])
        d_dc = torch.empty_like(fdc) if want_sh else None
        d_rest = torch.empty_like(frest) if (want_sh and frest is not None) else None
        zeros = lambda like: torch.zeros_like(like)
        if d_scaling is not None and g_scales is None:
            d_scaling = zeros(scaling)
        if d_rot is not None and g_rot is None:
            d_rot = zeros(rotation)
        if want_sh and g_shs is None:
            d_dc, d_rest = zeros(fdc), (zeros(frest) if frest is not None else None)
        if (d_scaling is not None and g_scales is not None) or (d_rot is not None and g_rot is not None) or \
                (want_sh and g_shs is not None):
            _lib.check(lib.c3dgs_qat_codebooks_backward(
                C.byref(q), _ptr(g_scales) if d_scaling is not None else None, _ptr(g_rot) if d_rot is not None else None,
                _ptr(g_shs) if want_sh else None, _ptr(d_scaling), _ptr(d_rot), _ptr(d_dc), _ptr(d_rest), s))
        return (None, None, None, None, d_xyz, d_screen, d_op, d_sf, d_scaling, d_rot,
                d_dc if need[10] else None, d_rest if need[11] else None)


class PipelineParams:
    """The three switches GaussianModel.render reads from the reference's arguments.PipelineParams."""

    def __init__(self, convert_SHs_python=False, compute_cov3D_python=False, debug=False):
        self.convert_SHs_python = convert_SHs_python
        self.compute_cov3D_python = compute_cov3D_python
        self.debug = debug


class GaussianModel:
    """Render-path mirror of scene/gaussian_model.py:GaussianModel (same constructor arguments, parameter attribute
    names, getters and render()). Parameters are plain tensors the caller assigns (`set_tensors`)."""

    def __init__(self, sh_degree, quantization=True, use_factor_scaling=True, device="cuda", is_splitted=True):
        self.is_splitted = is_splitted
        self.device = torch.device(device)
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.quantization = quantization
        self.use_factor_scaling = use_factor_scaling
        self.color_index_mode = ColorMode.NOT_INDEXED
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = None
        self._scaling_factor = None
        self._feature_indices = self._gaussian_indices = None
        self._fq_state = new_fq_state(self.device, len(SLOTS))
        self._modules_qa = {k: FakeQuantize(_row=self._fq_state[i]) for i, k in enumerate(SLOTS)}
        self.opacity_qa = self._modules_qa["opacity"]
        self.scaling_qa = self._modules_qa["scaling"]
        self.scaling_factor_qa = self._modules_qa["scaling_factor"]
        self.rotation_qa = self._modules_qa["rotation"]
        self.features_dc_qa = self._modules_qa["features_dc"]
        self.features_rest_qa = self._modules_qa["features_rest"]
        self.xyz_qa = FakeQuantizationHalf.apply
        if not quantization:                                    # gaussian_model.py:120-134
            for k in ("features_dc", "features_rest", "scaling", "scaling_factor", "rotation"):
                self._modules_qa[k].disable_fake_quant()
                self._modules_qa[k].disable_observer()
            self.xyz_qa = lambda x: x
        self._count_host = None
        self.spatial_lr_scale = 0.0                             # set by the scene loader in the reference (gaussian_model.py:286)
        self.optimizer = None
        self.xyz_scheduler_args = None
        # activations, gaussian_model.py:54-77
        if use_factor_scaling:
            self.scaling_activation = lambda x: torch.nn.functional.normalize(torch.nn.functional.relu(x))
            self.scaling_factor_activation = torch.exp
        else:
            self.scaling_activation = torch.exp
        self.opacity_activation = torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    # ---- parameters
    def set_tensors(self, *, xyz, features_dc, features_rest, scaling, rotation, opacity, scaling_factor=None,
                    feature_indices=None, gaussian_indices=None, active_sh_degree=None, requires_grad=True):
        """Install the learnable tensors (shapes as in the reference: xyz [P,3], features_dc [S,1,3], features_rest
        [S,M-1,3], scaling [G,3], rotation [G,4], opacity [P,1], scaling_factor [P,1]; S = G = P when not indexed)."""
        def param(t):
            return None if t is None else t.detach().to(self.device, torch.float32).contiguous().requires_grad_(requires_grad)
        self._xyz, self._features_dc, self._features_rest = param(xyz), param(features_dc), param(features_rest)
        self._scaling, self._rotation, self._opacity = param(scaling), param(rotation), param(opacity)
        self._scaling_factor = param(scaling_factor) if self.use_factor_scaling else None
        if self.use_factor_scaling and self._scaling_factor is None:
            raise RuntimeError("use_factor_scaling=True needs scaling_factor")
        self._feature_indices = None if feature_indices is None else feature_indices.to(self.device, torch.int64).contiguous()
        self._gaussian_indices = None if gaussian_indices is None else gaussian_indices.to(self.device, torch.int64).contiguous()
        self.color_index_mode = ColorMode.ALL_INDEXED if feature_indices is not None else ColorMode.NOT_INDEXED
        self.active_sh_degree = self.max_sh_degree if active_sh_degree is None else active_sh_degree
        return self

    def parameters(self):
        return [t for t in (self._xyz, self._features_dc, self._features_rest, self._scaling, self._scaling_factor,
                            self._rotation, self._opacity) if t is not None]

    @property
    def is_gaussian_indexed(self):
        return self._gaussian_indices is not None

    @property
    def is_color_indexed(self):
        return self._feature_indices is not None

    # ---- getters (gaussian_model.py:213-267). Each access runs its module's observer once, as in the reference.
    def _get(self, **raw):
        return _QatGetters.apply(self, None, None, None, raw.get("xyz"), None, raw.get("opacity"), raw.get("scaling_factor"),
                                 raw.get("scaling"), raw.get("rotation"), raw.get("fdc"), raw.get("frest"))

    @property
    def get_xyz(self):
        return self.xyz_qa(self._xyz)

    @property
    def get_opacity(self):
        return self._get(opacity=self._opacity)[2]

    @property
    def get_scaling_normalized(self):
        if self.use_factor_scaling:
            return self._get(scaling=self._scaling)[4]
        return self.scaling_qa(self.scaling_activation(self._scaling))

    @property
    def get_scaling_factor(self):
        if self._scaling_factor is None:
            return 1.0
        return self._get(scaling_factor=self._scaling_factor)[3]

    @property
    def get_scaling(self):
        scaling_n = self.get_scaling_normalized
        if self._scaling_factor is None:
            return scaling_n
        factor = self.get_scaling_factor
        return factor * (scaling_n[self._gaussian_indices] if self.is_gaussian_indexed else scaling_n)

    @property
    def _rotation_post_activation(self):
        return self._get(rotation=self._rotation)[5]

    @property
    def get_rotation(self):
        r = self._rotation_post_activation
        return r[self._gaussian_indices] if self.is_gaussian_indexed else r

    @property
    def _get_features_raw(self):
        return self._get(fdc=self._features_dc, frest=self._features_rest)[6]

    @property
    def get_features(self):
        f = self._get_features_raw
        return f[self._feature_indices] if self.color_index_mode == ColorMode.ALL_INDEXED else f

    def get_covariance(self, scaling_modifier=1, strip_sym=True):
        return _covariance(self.get_scaling, scaling_modifier, self.get_rotation, strip_sym)

    def get_normalized_covariance(self, scaling_modifier=1, strip_sym=True):
        return _covariance(self.get_scaling_normalized, scaling_modifier, self.get_rotation, strip_sym)

    # ---- what compress_gaussians needs (gaussian_model.py:1027-1059)
    def mask_splats(self, mask):
        with torch.no_grad():
            keep = lambda t: None if t is None else t.detach()[mask].contiguous().requires_grad_(t.requires_grad)
            self._xyz, self._opacity, self._scaling_factor = keep(self._xyz), keep(self._opacity), keep(self._scaling_factor)
            if self.is_color_indexed:
                self._feature_indices = self._feature_indices[mask].contiguous()
            else:
                self._features_dc, self._features_rest = keep(self._features_dc), keep(self._features_rest)
            if self.is_gaussian_indexed:
                self._gaussian_indices = self._gaussian_indices[mask].contiguous()
            else:
                self._scaling, self._rotation = keep(self._scaling), keep(self._rotation)

    def set_color_indexed(self, features, indices):
        self._feature_indices = indices.detach().to(self.device, torch.int64).contiguous()
        self._features_dc = features[:, :1].detach().contiguous().requires_grad_(True)
        self._features_rest = features[:, 1:].detach().contiguous().requires_grad_(True)
        self.color_index_mode = ColorMode.ALL_INDEXED

    def set_gaussian_indexed(self, rotation, scaling, indices):
        self._gaussian_indices = indices.detach().to(self.device, torch.int64).contiguous()
        self._rotation = rotation.detach().contiguous().requires_grad_(True)
        self._scaling = scaling.detach().contiguous().requires_grad_(True)

    def zero_grad(self):
        for t in self.parameters():
            t.grad = None

    # ---- on-disk payload (gaussian_model.py:505-623 save_npz, :625-720 load_npz, :997-1023 _sort_morton)
    def _sort_morton(self):
        from . import encode
        with torch.no_grad():
            order = encode.morton_order(self._xyz.detach())
            take = lambda t: None if t is None else t.detach()[order].contiguous().requires_grad_(t.requires_grad)
            self._xyz, self._opacity, self._scaling_factor = take(self._xyz), take(self._opacity), take(self._scaling_factor)
            if self.is_color_indexed:
                self._feature_indices = self._feature_indices[order].contiguous()
            else:
                self._features_rest, self._features_dc = take(self._features_rest), take(self._features_dc)
            if self.is_gaussian_indexed:
                self._gaussian_indices = self._gaussian_indices[order].contiguous()
            else:
                self._scaling, self._rotation = take(self._scaling), take(self._rotation)

    def quantized_payload(self):
        """int8 codes of every fake-quantised tensor with the modules' current scale / zero_point, as
        torch.quantize_per_tensor(...).int_repr() gives them in save_npz: ONE launch for all six tensors."""
        dev = self.device
        i8 = lambda t: None if t is None else torch.empty(t.shape, dtype=torch.int8, device=dev)
        raw = dict(opacity=self._opacity, scaling=self._scaling, scaling_factor=self._scaling_factor, rotation=self._rotation,
                   fdc=self._features_dc, frest=self._features_rest)
        raw = {k: (None if v is None else v.detach().contiguous()) for k, v in raw.items()}
        out = {k: i8(v) for k, v in raw.items()}
        q = _qat_params(self, self._fq_state, **raw)
        _lib.check(_lib.lib().c3dgs_qat_quantize(C.byref(q), int(not self.use_factor_scaling), _ptr(out["opacity"]),
                                                 _ptr(out["scaling"]), _ptr(out["scaling_factor"]), _ptr(out["rotation"]),
                                                 _ptr(out["fdc"]), _ptr(out["frest"]), _stream(dev)))
        return out

    def save_npz(self, path, compress=True, half_precision=False, sort_morton=False):
        """Same keys, dtypes and shapes as the reference writes, so its load_npz / the web viewer read the file."""
        import os
        import numpy as np
        with torch.no_grad():
            if sort_morton:
                self._sort_morton()
            if isinstance(path, str):
                os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            dtype = torch.half if half_precision else torch.float32
            d = {"quantization": self.quantization}
            host = lambda t: t.detach().cpu().numpy()
            if self.quantization:
                codes = self.quantized_payload()
                d["xyz"] = host(self._xyz.detach().half())
                for key, name, slot in (("features_dc", "fdc", "features_dc"), ("features_rest", "frest", "features_rest"),
                                        ("opacity", "opacity", "opacity"), ("scaling", "scaling", "scaling"),
                                        ("scaling_factor", "scaling_factor", "scaling_factor"),
                                        ("rotation", "rotation", "rotation")):
                    if codes[name] is None:
                        continue
                    mod = self._modules_qa[slot]
                    d[key] = host(codes[name])
                    d[key + "_scale"] = host(mod.scale)
                    d[key + "_zero_point"] = host(mod.zero_point)
            else:
                d["xyz"] = host(self._xyz)
                d["features_dc"], d["features_rest"] = host(self._features_dc), host(self._features_rest)
                d["opacity"] = host(self._opacity.detach().to(dtype))
                d["scaling"] = host(self._scaling.detach().to(dtype))
                if self._scaling_factor is not None:
                    d["scaling_factor"] = host(self._scaling_factor.detach().to(dtype))
                d["rotation"] = host(self._rotation.detach().to(dtype))
            if self.is_color_indexed:
                d["feature_indices"] = host(self._feature_indices.int())
            if self.is_gaussian_indexed:
                d["gaussian_indices"] = host(self._gaussian_indices.int())
            # key order of the reference's file
            order = ["quantization", "xyz", "features_dc", "features_dc_scale", "features_dc_zero_point", "features_rest",
                     "features_rest_scale", "features_rest_zero_point", "opacity", "opacity_scale", "opacity_zero_point",
                     "feature_indices", "gaussian_indices", "scaling", "scaling_scale", "scaling_zero_point",
                     "scaling_factor", "scaling_factor_scale", "scaling_factor_zero_point", "rotation", "rotation_scale",
                     "rotation_zero_point"]
            (np.savez_compressed if compress else np.savez)(path, **{k: d[k] for k in order if k in d})

    def load_npz(self, path, override_quantization=False):
        import numpy as np
        sd = np.load(path)                                      # plain arrays only (allow_pickle stays False)
        quantization = bool(sd["quantization"])
        if not override_quantization and self.quantization != quantization:
            print("WARNING: model is not quantisation aware but loaded model is")
        if override_quantization:
            self.quantization = quantization
        dev = self.device
        par = lambda t: t.to(dev, torch.float32).contiguous().requires_grad_(True)

        def dequant(key, slot):
            qv = torch.from_numpy(sd[key]).int().to(dev)
            scale = torch.from_numpy(sd[key + "_scale"]).to(dev)
            zp = torch.from_numpy(sd[key + "_zero_point"]).to(dev)
            val = (qv - zp) * scale
            row = self._modules_qa[slot]._row
            row[0], row[1], row[2] = val.min(), val.max(), scale.reshape(-1)[0]
            row[3:4].view(torch.int32)[0] = zp.reshape(-1)[0].int()
            return val

        self._xyz = par(torch.from_numpy(sd["xyz"]).float())
        if quantization:
            self._features_rest = par(dequant("features_rest", "features_rest"))
            self._features_dc = par(dequant("features_dc", "features_dc"))
            op = dequant("opacity", "opacity")
            self._opacity = par(torch.log(op / (1 - op)))       # inverse_sigmoid
            sc = dequant("scaling", "scaling")
            self._scaling = par(sc if self.use_factor_scaling else torch.log(sc))
            self._scaling_factor = par(dequant("scaling_factor", "scaling_factor")) if "scaling_factor" in sd else None
            self._rotation = par(dequant("rotation", "rotation"))
        else:
            self._features_dc, self._features_rest = par(torch.from_numpy(sd["features_dc"]).float()), \
                par(torch.from_numpy(sd["features_rest"]).float())
            self._opacity = par(torch.from_numpy(sd["opacity"]).float())
            self._scaling_factor = par(torch.from_numpy(sd["scaling_factor"]).float()) if "scaling_factor" in sd else None
            self._scaling = par(torch.from_numpy(sd["scaling"]).float())
            self._rotation = par(torch.from_numpy(sd["rotation"]).float())
        self._feature_indices = torch.from_numpy(sd["feature_indices"]).long().to(dev) if "feature_indices" in sd else None
        self._gaussian_indices = torch.from_numpy(sd["gaussian_indices"]).long().to(dev) if "gaussian_indices" in sd else None
        self.color_index_mode = ColorMode.ALL_INDEXED if self._feature_indices is not None else ColorMode.NOT_INDEXED
        self.active_sh_degree = self.max_sh_degree
        return self

    # ---- optimizer plumbing of the fine-tuning loop (gaussian_model.py:292-322)
    def training_setup(self, training_args):
        """Same parameter groups, names and learning rates as the reference; the optimizer is the fused Adam
        (c3dgs_amd.optim.Adam, one launch for all groups) instead of torch.optim.Adam(l, lr=0.0, eps=1e-15)."""
        from . import optim
        self.percent_dense = training_args.percent_dense
        n = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((n, 1), device=self.device)
        self.denom = torch.zeros((n, 1), device=self.device)
        groups = [
            {"params": [self._xyz], "lr": training_args.position_lr_init * self.spatial_lr_scale, "name": "xyz"},
            {"params": [self._features_dc], "lr": training_args.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": training_args.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": training_args.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": training_args.scaling_lr, "name": "scaling"},
            {"params": [self._rotation], "lr": training_args.rotation_lr, "name": "rotation"},
        ]
        if self._scaling_factor is not None:
            groups.append({"params": [self._scaling_factor], "lr": training_args.scaling_lr, "name": "scaling_factor"})
        self.optimizer = optim.Adam(groups, lr=0.0, eps=1e-15)
        self.xyz_scheduler_args = get_expon_lr_func(
            lr_init=training_args.position_lr_init * self.spatial_lr_scale,
            lr_final=training_args.position_lr_final * self.spatial_lr_scale,
            lr_delay_mult=training_args.position_lr_delay_mult, max_steps=training_args.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        for param_group in self.optimizer.param_groups:
            if param_group["name"] == "xyz":
                lr = self.xyz_scheduler_args(iteration)
                param_group["lr"] = lr
                return lr

    # ---- render (gaussian_model.py:766-886)
    def render(self, viewpoint_camera, pipe, bg_color, scaling_modifier=1.0, override_color=None, clamp_color=True,
               cov3d=None, gather_visible=True):
        if pipe.convert_SHs_python and override_color is None:
            raise NotImplementedError("convert_SHs_python: SH evaluation in Python is outside the mirrored render path")
        dev = self.device
        settings = GaussianRasterizationSettings(
            intrinsic=viewpoint_camera.intrinsic.to(dev), extrinsic_vector=viewpoint_camera.extrinsic_vector.to(dev),
            bg=bg_color.to(dev), scale_modifier=scaling_modifier, sh_degree=self.active_sh_degree, prefiltered=False,
            debug=pipe.debug, clamp_color=clamp_color)
        indexed = self.color_index_mode == ColorMode.ALL_INDEXED and self.is_gaussian_indexed
        fused = indexed and self.use_factor_scaling and cov3d is None and override_color is None and \
            not pipe.compute_cov3D_python
        screenspace_points = torch.zeros(self._xyz.shape, dtype=torch.float32, device=dev, requires_grad=True)
        if fused:
            return self._render_indexed_fused(settings, screenspace_points)
        return self._render_composed(settings, screenspace_points, indexed, pipe, scaling_modifier, override_color, cov3d,
                                     gather_visible)

    def _visible(self, settings):
        """visible flags, their exclusive scan and a deferred host read of the count."""
        lib = _lib.lib()
        dev = self.device
        P = self._xyz.shape[0]
        view = _rz.camera_matrices(settings.intrinsic, settings.extrinsic_vector, dev)[0]
        visible = torch.empty(P, dtype=torch.uint8, device=dev)
        rank = torch.empty(P, dtype=torch.int32, device=dev)
        count = torch.empty(1, dtype=torch.int32, device=dev)
        scan = torch.empty(max(int(lib.c3dgs_qat_scan_bytes(P)), 256), dtype=torch.uint8, device=dev)
        q = _qat_params(self, self._fq_state, xyz=self._xyz)
        _lib.check(lib.c3dgs_qat_visible(C.byref(q), view.data_ptr(), visible.data_ptr(), rank.data_ptr(), count.data_ptr(),
                                         scan.data_ptr(), _stream(dev)))
        if self._count_host is None:
            self._count_host = torch.empty(1, dtype=torch.int32).pin_memory()
        self._count_host.copy_(count, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))

        def read():
            spins = 0
            while not ev.query():                 # poll: a blocking wait can cost milliseconds of wake-up on a busy host
                spins += 1
                if spins > 2_000_000:
                    ev.synchronize()
                    break
            return int(self._count_host[0])
        return visible, rank, read

    def _render_indexed_fused(self, settings, screenspace_points):
        visible, rank, read = self._visible(settings)
        (means3D, means2D, opac, sfac, scales_n, rotations, shs, sh_idx, g_idx) = _QatGetters.apply(
            self, (visible, rank, read), self._feature_indices, self._gaussian_indices, self._xyz, screenspace_points,
            self._opacity, self._scaling_factor, self._scaling, self._rotation, self._features_dc, self._features_rest)
        # what GaussianRasterizerIndexed(settings, optimize_camera=True)(...) calls, without building an nn.Module per view
        # (the host has ~0.2 ms of Python between the visible count's arrival and the rasterizer's first launch, and the GPU
        # idles for the part of it the getter kernels do not cover)
        image, radii = _rz.rasterize_gaussians_indexed_camera(means3D, means2D, shs, sh_idx, g_idx, _rz._empty(), opac, scales_n, sfac,
                                                              rotations, _rz._empty(), settings, settings.extrinsic_vector)
        return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
                "visible": visible.bool()}

    def _render_composed(self, settings, screenspace_points, indexed, pipe, scaling_modifier, override_color, cov3d,
                         gather_visible=True):
        """Every other configuration of render(): the same getters, composed with torch gathers like the reference.
        gather_visible=False (the sensitivity pass) hands all rows to the rasterizer instead of the reference's `t[visible]`
        copies: the rasterizer culls the same Gaussians itself (mark_visible is its own frustum test), so image and gradients
        are the same, `radii` / `viewspace_points` are then indexed by Gaussian rather than by visible row."""
        means3D = self.get_xyz
        opacity = self.get_opacity
        rasterizer = GaussianRasterizerIndexed(raster_settings=settings, optimize_camera=True) if indexed \
            else GaussianRasterizer(raster_settings=settings)
        scales = rotations = None
        cov3D_precomp = cov3d
        if cov3D_precomp is None:
            if pipe.compute_cov3D_python:
                cov3D_precomp = self.get_covariance(scaling_modifier)
            else:
                scales = self.get_scaling_normalized if indexed else self.get_scaling
                rotations = self._rotation_post_activation if indexed else self.get_rotation
        scale_factors = self.get_scaling_factor if indexed else None
        shs = colors_precomp = None
        if override_color is None:
            shs = self._get_features_raw if indexed else self.get_features
        else:
            colors_precomp = override_color
        visible = rasterizer.markVisible(means3D, extrinsic_vector=settings.extrinsic_vector)
        # `t[visible]` of the reference (gaussian_model.py:851-862) for every input, from ONE nonzero: a boolean-mask index
        # runs nonzero (with its host sync) per tensor, and its backward is a sort-based index_put(accumulate) -- 2 ms
        # per tensor for 6M rows -- although the rows are unique.
        if gather_visible:
            rows = visible.nonzero(as_tuple=False).squeeze(1)
            pick = lambda t: None if t is None else (_MaskGather.apply(t, rows) if t.requires_grad else t.index_select(0, rows))  # noqa: E731
        else:
            pick = lambda t: t  # noqa: E731
        if indexed:
            image, radii = rasterizer(means3D=pick(means3D), means2D=pick(screenspace_points), shs=shs,
                                      sh_indices=pick(self._feature_indices), g_indices=pick(self._gaussian_indices),
                                      colors_precomp=None, opacities=pick(opacity), scales=scales,
                                      scale_factors=pick(scale_factors), rotations=rotations,
                                      cov3D_precomp=pick(cov3D_precomp), extrinsic_vector=settings.extrinsic_vector)
        else:
            image, radii = rasterizer(means3D=pick(means3D), means2D=pick(screenspace_points), shs=pick(shs),
                                      colors_precomp=pick(colors_precomp), opacities=pick(opacity), scales=pick(scales),
                                      rotations=pick(rotations), cov3D_precomp=pick(cov3D_precomp),
                                      extrinsic_vector=settings.extrinsic_vector)
        return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
                "visible": visible}


class _MaskGather(torch.autograd.Function):
    """t[mask] with the row numbers of the mask given: forward index_select, backward a plain scatter into zeros (the rows
    are unique, nothing accumulates)."""

    @staticmethod
    def forward(ctx, t, rows):
        ctx.save_for_backward(rows)
        ctx.n = t.shape[0]
        return t.index_select(0, rows)

    @staticmethod
    def backward(ctx, g):
        (rows,) = ctx.saved_tensors
        out = g.new_zeros((ctx.n,) + tuple(g.shape[1:]))
        out.index_copy_(0, rows, g.contiguous())
        return out, None


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """utils/general_utils.py:32-65: log-linear interpolation from lr_init (step 0) to lr_final (step max_steps), optionally
    eased in over lr_delay_steps; 0 for negative steps or when both rates are 0."""
    import math

    def rate(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        delay = 1.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        t = min(max(step / max_steps, 0.0), 1.0)
        return delay * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
    return rate


def _covariance(scaling, scaling_modifier, rotation, strip_sym=True):
    """build_covariance_from_scaling_rotation (gaussian_model.py:55-64): Sigma = R S S^T R^T, upper triangle."""
    r = rotation / rotation.norm(dim=1, keepdim=True)          # build_rotation normalises again (general_utils.py:84-89)
    w, x, y, z = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    L = R * (scaling_modifier * scaling)[:, None, :]
    # L @ L^T written out: a batched 3x3 GEMM of millions of matrices runs at a few GB/s in the BLAS library (52 ms for 6M)
    r0, r1, r2 = L[:, 0], L[:, 1], L[:, 2]
    sym = torch.stack([(r0 * r0).sum(1), (r0 * r1).sum(1), (r0 * r2).sum(1), (r1 * r1).sum(1), (r1 * r2).sum(1),
                       (r2 * r2).sum(1)], dim=1)
    if strip_sym:
        return sym
    return sym[:, [0, 1, 2, 1, 3, 4, 2, 4, 5]].reshape(-1, 3, 3)
