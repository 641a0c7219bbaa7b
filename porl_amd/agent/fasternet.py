"""FasterNet costmap encoder — drop-in for /root/reference/agent/fasternet.py:264-438 as SORL uses it
(`FasterNet(3, args.feature_dim)`, sorl_train.py:29; `forward_cls`, :428-438) on one MI355X.

The module tree only HOLDS the parameters: it has the reference's names, shapes, registration order and
initialisation (so the same `torch.manual_seed` gives bit-identical weights and `state_dict()`s
interchange); every tensor is a view into one flat device buffer and the forward runs in
libporl_hip.so (`porl_enc_forward`): state2costmap -> 4x4 patch embedding -> BatchNorm -> MLPBlocks
(3x3 partial conv + two 1x1 convs as fp32-MFMA GEMMs over NHWC position rows) -> 2x2 merge -> MLPBlocks ->
global average pool -> 1x1 conv + ReLU -> Linear.

Forward only.  The reference keeps the backbone out of both optimizers (agent/sorl.py:58-64): its backward
pass only fills `.grad` fields nobody reads, so skipping it changes no result.  Consequently the parameters
here do not require grad.

Train mode (the reference never calls `.eval()`): BatchNorm normalises with batch statistics and updates the
running ones; DropPath (fasternet.py:76-93) draws one Bernoulli keep flag per sample and block.  The flags
are drawn on the HOST from torch's default CPU generator, in the reference's order, so a CPU reference with
the same seed sees the same masks.  Like the reference, `forward` zeroes entries > 8 of its input in place
(util/costmap.py:17).
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from .. import _native as N
from ..engine import _norm_device


class _Identity(nn.Identity):
    drop_prob = 0.0


class DropPath(nn.Module):
    """Parameter-free marker; the keep mask is applied inside the engine (residual_kernel)."""

    def __init__(self, drop_prob=0., scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def extra_repr(self):
        return f'drop_prob={round(self.drop_prob, 3):0.3f}'


class Partial_conv3(nn.Module):
    def __init__(self, dim, n_div):
        super().__init__()
        self.dim_conv3 = dim // n_div
        self.dim_untouched = dim - self.dim_conv3
        self.partial_conv3 = nn.Conv2d(self.dim_conv3, self.dim_conv3, 3, 1, 1, bias=False)


class MLPBlock(nn.Module):
    def __init__(self, dim, n_div, mlp_ratio, drop_path):
        super().__init__()
        self.dim, self.mlp_ratio, self.n_div = dim, mlp_ratio, n_div
        self.drop_path = DropPath(drop_path) if drop_path > 0. else _Identity()
        hidden = int(dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Conv2d(dim, hidden, 1, bias=False), nn.BatchNorm2d(hidden), nn.ReLU(inplace=True),
                                 nn.Conv2d(hidden, dim, 1, bias=False))
        self.spatial_mixing = Partial_conv3(dim, n_div)


class BasicStage(nn.Module):
    def __init__(self, dim, depth, n_div, mlp_ratio, drop_path):
        super().__init__()
        self.blocks = nn.Sequential(*[MLPBlock(dim, n_div, mlp_ratio, drop_path[i]) for i in range(depth)])


class PatchEmbed(nn.Module):
    def __init__(self, patch_size, patch_stride, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_stride, bias=False)
        self.norm = nn.BatchNorm2d(embed_dim)


class PatchMerging(nn.Module):
    def __init__(self, patch_size2, patch_stride2, dim):
        super().__init__()
        self.reduction = nn.Conv2d(dim, 2 * dim, kernel_size=patch_size2, stride=patch_stride2, bias=False)
        self.norm = nn.BatchNorm2d(2 * dim)


class FasterNet(nn.Module):
    ANGLE_BINS, DIST_BINS = 360, 256          # util/costmap.py:7 defaults, the image forward_cls builds

    def __init__(self, in_chans=3, num_classes=1000, embed_dim=96, depths=(1, 2), mlp_ratio=2., n_div=4,
                 patch_size=4, patch_stride=4, patch_size2=2, patch_stride2=2, patch_norm=True, feature_dim=1280,
                 drop_path_rate=0.1, layer_scale_init_value=0, norm_layer='BN', act_layer='RELU', fork_feat=False,
                 init_cfg=None, pretrained=None, pconv_fw_type='split_cat', max_batch=512, angle_bins=None,
                 dist_bins=None, compute_dtype="fp32", **kwargs):
        super().__init__()
        # costmap geometry: the reference rasterises to 360 x 256 (util/costmap.py:7,12,24) and cannot do otherwise;
        # `angle_bins` / `dist_bins` (multiples of 4) parametrise it, e.g. the 84 x 84 image BASELINE config 5 names.
        # The state then carries angle_bins beams + the 2 goal coordinates.
        # compute_dtype="bf16" (BASELINE config 5's wording; the reference is fp32 everywhere): the backbone after the patch
        # embedding keeps its activations in HBM as bf16 and runs the partial 3x3 conv, the MLP blocks (one kernel per pass
        # over x: W1 -> BatchNorm -> ReLU -> W2 -> residual) and the 2x2s2 merge on the bf16 matrix pipe with fp32
        # accumulation (csrc/encoder_bf16.hpp); BatchNorm statistics, the patch embedding's arithmetic and the pooled head
        # stay fp32.  Shapes other than the reference architecture (embed 96, mlp_ratio 2, n_div 4) fall back to
        # "bf16_operands": fp32 tensors in memory, operands rounded to bf16 on their way into LDS (round 2's mode, still
        # selectable by name).  fp32 is the default and the parity path.
        if compute_dtype not in ("fp32", "bf16", "bf16_operands"):
            raise ValueError("compute_dtype must be 'fp32', 'bf16' or 'bf16_operands'")
        self.compute_dtype = compute_dtype
        if angle_bins is not None:
            self.ANGLE_BINS = int(angle_bins)
        if dist_bins is not None:
            self.DIST_BINS = int(dist_bins)
        if norm_layer != 'BN' or act_layer != 'RELU':
            raise NotImplementedError("only norm_layer='BN', act_layer='RELU' (the configuration sorl_train.py builds)")
        if fork_feat or layer_scale_init_value > 0 or not patch_norm or pretrained is not None or init_cfg is not None:
            raise NotImplementedError("fork_feat / layer_scale / patch_norm=False / pretrained weights are outside the path")
        if in_chans != 3 or (patch_size, patch_stride, patch_size2, patch_stride2) != (4, 4, 2, 2) or len(depths) != 2:
            raise NotImplementedError("the costmap encoder is 3 channels, 4x4s4 patches, one 2x2s2 merge, two stages")
        if pconv_fw_type not in ('split_cat', 'slicing'):
            raise NotImplementedError
        if num_classes <= 0:
            raise NotImplementedError("num_classes must be positive (Linear head)")
        self.num_classes = num_classes
        self.num_stages = len(depths)
        self.embed_dim = embed_dim
        self.patch_norm = patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_stages - 1))
        self.mlp_ratio = mlp_ratio
        self.depths = tuple(depths)
        self.feature_dim = feature_dim
        self.fork_feat = False

        # construction order == the reference's, so default initialisers consume the generator identically
        self.patch_embed = PatchEmbed(patch_size, patch_stride, in_chans, embed_dim)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        stages = []
        for i_stage in range(self.num_stages):
            stages.append(BasicStage(int(embed_dim * 2 ** i_stage), depths[i_stage], n_div, mlp_ratio,
                                     dpr[sum(depths[:i_stage]):sum(depths[:i_stage + 1])]))
            if i_stage < self.num_stages - 1:
                stages.append(PatchMerging(patch_size2, patch_stride2, int(embed_dim * 2 ** i_stage)))
        self.stages = nn.Sequential(*stages)
        self.avgpool_pre_head = nn.Sequential(nn.AdaptiveAvgPool2d(1),
                                              nn.Conv2d(self.num_features, feature_dim, 1, bias=False),
                                              nn.ReLU(inplace=True))
        self.head = nn.Linear(feature_dim, num_classes)
        self.apply(self.cls_init_weights)
        self._drop_probs = [float(p) for p in dpr]

        self._lib = N.lib()
        self._cfg = N.EncCfg(self.ANGLE_BINS, self.DIST_BINS, int(embed_dim), int(depths[0]), int(depths[1]),
                             int(n_div), int(feature_dim), int(num_classes), int(max_batch), float(mlp_ratio),
                             float(self.patch_embed.norm.eps), float(self.patch_embed.norm.momentum),
                             {"fp32": 0, "bf16_operands": 1, "bf16": 2}[compute_dtype])
        h = C.c_void_p()
        N.check(self._lib.porl_enc_create(C.byref(self._cfg), C.byref(h)), "porl_enc_create")
        self._h = h
        self._device = torch.device("cpu")
        self._flat = torch.zeros(int(self._lib.porl_enc_param_floats(h)), dtype=torch.float32)
        self._stats = torch.zeros(int(self._lib.porl_enc_stat_floats(h)), dtype=torch.float32)
        self._workspace = None
        self._bound = False
        self._adopt(copy_from_modules=True)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.weights_changed())

    def weights_changed(self):
        """The engine keeps permuted copies of the convolution weights; tell it the parameters were rewritten
        (called by load_state_dict and .to(); call it yourself after editing parameters in place)."""
        N.check(self._lib.porl_enc_weights_changed(self._h), "porl_enc_weights_changed")

    # reference fasternet.py:380-390
    @staticmethod
    def cls_init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.Conv1d, nn.Conv2d)):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    # -- flat storage ------------------------------------------------------------------------------
    def _tables(self):
        off, numel, ch, off2 = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int64()
        name = C.create_string_buffer(128)
        params, norms = {}, {}
        for i in range(int(self._lib.porl_enc_tensors(self._h))):
            N.check(self._lib.porl_enc_tensor_info(self._h, i, C.byref(off), C.byref(numel), name, 128))
            params[name.value.decode()] = (off.value, numel.value)
        for i in range(int(self._lib.porl_enc_norms(self._h))):
            N.check(self._lib.porl_enc_norm_info(self._h, i, C.byref(off), C.byref(off2), C.byref(ch), name, 128))
            norms[name.value.decode()] = (off.value, off2.value, ch.value)
        return params, norms

    def _adopt(self, copy_from_modules=False):
        """Point every parameter / BatchNorm running statistic at its view in the flat buffers."""
        params, norms = self._tables()
        named = dict(nn.Module.named_parameters(self))
        if set(named) != set(params):
            raise RuntimeError(f"engine/module parameter names differ: {sorted(set(named) ^ set(params))}")
        with torch.no_grad():
            for key, p in named.items():
                o, n = params[key]
                if n != p.numel():
                    raise RuntimeError(f"{key}: {p.numel()} elements vs engine {n}")
                v = self._flat[o:o + n].view(p.shape)
                if copy_from_modules:
                    v.copy_(p)
                p.data = v
                p.requires_grad_(False)
            self._nbt = []
            for prefix, (om, ov, ch) in norms.items():
                bn = self.get_submodule(prefix)
                vm, vv = self._stats[om:om + ch], self._stats[ov:ov + ch]
                if copy_from_modules:
                    vm.copy_(bn.running_mean)
                    vv.copy_(bn.running_var)
                bn.running_mean, bn.running_var = vm, vv
                bn.num_batches_tracked = bn.num_batches_tracked.to(self._device)
                self._nbt.append(bn.num_batches_tracked)

    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0, dtype=torch.float32, device=self._device))
        if probe.dtype != torch.float32:
            raise RuntimeError("the encoder is fp32 only")
        dev = _norm_device(probe.device)
        if dev != self._device:
            self._flat, self._stats = self._flat.to(dev), self._stats.to(dev)
            self._device = dev
            self._workspace, self._bound = None, False
            self._adopt()
        return self

    def _ensure_bound(self):
        if self._device.type != "cuda":
            raise N.NativeError("the encoder computes on a HIP device only (.to('cuda')); there is no CPU path")
        if not self._bound:
            self._workspace = torch.empty(int(self._lib.porl_enc_workspace_floats(self._h)), dtype=torch.float32,
                                          device=self._device)
            N.check(self._lib.porl_enc_bind(self._h, N.ptr(self._flat), N.ptr(self._stats), N.ptr(self._workspace)),
                    "porl_enc_bind")
            self._bound = True

    # -- forward -----------------------------------------------------------------------------------
    def draw_drop_scale(self, batch):
        """(blocks, batch) DropPath factors, drawn like fasternet.py:86-93 block by block from the default CPU
        generator; None when no block drops (eval mode or drop_path_rate == 0)."""
        if not self.training or not any(p > 0. for p in self._drop_probs):
            return None
        rows = []
        for p in self._drop_probs:
            if p > 0.:
                keep = 1 - p
                r = torch.empty(batch, 1, 1, 1).bernoulli_(keep)
                if keep > 0.0:
                    r.div_(keep)
                rows.append(r.view(batch))
            else:
                rows.append(torch.ones(batch))
        return torch.stack(rows)

    def forward(self, x, drop_scale=None):
        """x (b, 362) fp32 on the device -> (b, num_classes).  `drop_scale` overrides the drawn DropPath factors."""
        self._ensure_bound()
        if x.dim() != 2 or x.shape[1] != self.ANGLE_BINS + 2:
            raise RuntimeError(f"state: expected (b, {self.ANGLE_BINS + 2}), got {tuple(x.shape)}")
        if x.dtype != torch.float32 or x.stride(1) != 1:
            raise RuntimeError("state must be fp32 with unit column stride (it is modified in place)")
        if _norm_device(x.device) != self._device:
            raise RuntimeError(f"state is on {x.device}, the encoder on {self._device}")
        b = x.shape[0]
        if b > self._cfg.max_batch:
            raise RuntimeError(f"batch {b} > max_batch {self._cfg.max_batch} the workspace was sized for")
        if drop_scale is None:
            drop_scale = self.draw_drop_scale(b)
        if drop_scale is not None:
            drop_scale = drop_scale.to(device=self._device, dtype=torch.float32).contiguous()
            if tuple(drop_scale.shape) != (len(self._drop_probs), b):
                raise RuntimeError(f"drop_scale: expected {(len(self._drop_probs), b)}, got {tuple(drop_scale.shape)}")
        out = torch.empty(b, self.num_classes, dtype=torch.float32, device=self._device)
        N.check(self._lib.porl_enc_forward(self._h, N.ptr(x), x.stride(0), b, int(self.training), N.ptr(drop_scale),
                                           N.ptr(out), out.stride(0), N.current_stream_ptr(self._device)), "porl_enc_forward")
        if self.training:
            torch._foreach_add_(self._nbt, 1)
        return out

    forward_cls = forward

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.porl_enc_destroy(self._h)
                self._h = None
        except Exception:
            pass
