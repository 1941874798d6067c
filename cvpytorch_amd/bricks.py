"""Layer registries + Hip layers: the drop-in boundary of the reference's `src/models/bricks`.

The reference threads `conv_cfg / norm_cfg / act_cfg` dicts down to `ConvModule`
(src/models/bricks/conv_module.py:74-89,119-168), which resolves them through six mmcv-style
registries (src/models/bricks/registry.py:4-9; src/utils/registry.py:81-346) with
`build_conv_layer` (bricks/conv.py:14-46), `build_norm_layer` (bricks/norm.py:74-123),
`build_activation_layer` (bricks/activation.py:86-98), `build_upsample_layer`
(bricks/upsample.py:53-87) and `build_plugin_layer` (bricks/plugin.py:58-94).

This module mirrors that API (same registry names, same builder signatures and error behaviour) and
registers the MI355X-native layers under new type names:

    conv_cfg=dict(type='HipConv2d')  norm_cfg=dict(type='HipBN', momentum=0.03, eps=0.001)
    act_cfg=dict(type='HipSiLU')     plugin 'HipConvModule' (fused conv+BN+act, ConvModule's ctor)

plus `convert_to_hip(model)` — a module-swap pass (same idiom as
`nn.SyncBatchNorm.convert_sync_batchnorm`, trainer.py:127) for layers created outside the registry
(detects/yolov5_detect.py:25, heads/seg/base_seg_head.py:30, torchvision Bottleneck ...). Hip layers
subclass the torch layers they replace, so parameter names, `state_dict()` keys, weight-init loops
(`isinstance(m, nn.Conv2d)`) and the optimizer's no-decay rule (optimizers/__init__.py:37,45) are
unchanged. Their forward has no eager fallback: off-GPU it raises.
"""
import inspect
import logging
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.batchnorm import _BatchNorm

from . import lib as L
from . import ops


# ------------------------------------------------------------------------------------------------------
# Registry (API of src/utils/registry.py:81-346, the subset the bricks use)
# ------------------------------------------------------------------------------------------------------
class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return "%s(name=%s, items=%s)" % (self.__class__.__name__, self._name, sorted(self._module_dict))

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def _register_module(self, module, module_name=None, force=False):
        if not inspect.isclass(module) and not inspect.isfunction(module):
            raise TypeError("module must be a class or a function, but got %s" % type(module))
        if module_name is None:
            module_name = module.__name__
        if isinstance(module_name, str):
            module_name = [module_name]
        for name in module_name:
            if not force and name in self._module_dict:
                raise KeyError("%s is already registered in %s" % (name, self.name))
            self._module_dict[name] = module

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError("force must be a boolean, but got %s" % type(force))
        if not (name is None or isinstance(name, str) or (isinstance(name, (list, tuple)) and all(isinstance(n, str) for n in name))):
            raise TypeError("name must be either of None, an instance of str or a sequence of str, but got %s" % type(name))
        if module is not None:
            self._register_module(module=module, module_name=name, force=force)
            return module

        def _register(mod):
            self._register_module(module=mod, module_name=name, force=force)
            return mod

        return _register


def build_from_cfg(cfg, registry, default_args=None):
    """src/utils/registry.py:16-78."""
    if not isinstance(cfg, dict):
        raise TypeError("cfg must be a dict, but got %s" % type(cfg))
    if "type" not in cfg:
        if default_args is None or "type" not in default_args:
            raise KeyError('`cfg` or `default_args` must contain the key "type", but got %s\n%s' % (cfg, default_args))
    if not isinstance(registry, Registry):
        raise TypeError("registry must be a Registry object, but got %s" % type(registry))
    args = cfg.copy()
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError("%s is not in the %s registry" % (obj_type, registry.name))
    elif inspect.isclass(obj_type) or inspect.isfunction(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError("type must be a str or valid type, but got %s" % type(obj_type))
    return obj_cls(**args)


CONV_LAYERS = Registry("conv layer")
NORM_LAYERS = Registry("norm layer")
ACTIVATION_LAYERS = Registry("activation layer")
PADDING_LAYERS = Registry("padding layer")
UPSAMPLE_LAYERS = Registry("upsample layer")
PLUGIN_LAYERS = Registry("plugin layer")


def _act_id(m):
    """(act id, parameter) for a torch / Hip activation module, or None if not expressible."""
    if m is None:
        return L.ACT_NONE, 0.0
    if isinstance(m, (nn.SiLU, Swish)):
        return L.ACT_SILU, 0.0
    if isinstance(m, nn.LeakyReLU):
        return L.ACT_LEAKY, float(m.negative_slope)
    if isinstance(m, nn.ReLU):
        return L.ACT_RELU, 0.0
    if isinstance(m, nn.Sigmoid):
        return L.ACT_SIGMOID, 0.0
    if isinstance(m, nn.Hardswish):
        return L.ACT_HSWISH, 0.0
    return None


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


# ------------------------------------------------------------------------------------------------------
# Hip layers
# ------------------------------------------------------------------------------------------------------
def bn_tick(bn):
    """BatchNorm's `num_batches_tracked += 1` (torch/nn/modules/batchnorm.py). A train state that owns the step
    (arena.FlatTrainState) sets `bn._nbt_deferred` and bumps all counters with ONE multi-tensor add per step instead of
    one 4-us kernel per layer."""
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None and not getattr(bn, "_nbt_deferred", False):
        bn.num_batches_tracked.add_(1)


class HipConv2d(nn.Conv2d):
    """nn.Conv2d whose forward/backward run on libcvhip's MFMA implicit-GEMM kernels."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.padding_mode != "zeros":
            raise L.CvhipError("HipConv2d supports zero padding only")
        if isinstance(self.padding, str):
            raise L.CvhipError("HipConv2d needs numeric padding")
        self._hip_state = ops.ConvState()
        # master weights live in KRSC physical order (OIHW shape, channels_last strides): the operand packers
        # read them without a relayout and wgrad's KRSC output IS the .grad tensor (no clone in AccumulateGrad)
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def _effective(self, x):
        """Image inputs (fp32 NCHW, C % 8 != 0) are relaid out to bf16 NHWC with zero-padded channels by one
        libcvhip kernel; the weight keeps its real shape (the packers pad it: cvhip_conv_desc.c_valid)."""
        w = self.weight
        if (x.dim() == 4 and x.shape[1] == self.in_channels and self.in_channels % 8 != 0 and self.groups == 1
                and not x.requires_grad and x.dtype != ops.ACT_DTYPE):
            if ops._STEM_IMAGE and x.dtype == torch.float32 and self.in_channels <= 4 and x.is_cuda and x.is_contiguous():
                # round 4: an image stem reads the fp32 NCHW batch itself (ops.ConvBnAct decides per descriptor and falls back to the
                # conversion pass below for everything the image-stem kernel does not run)
                return x, w
            cp = (self.in_channels + 7) // 8 * 8
            if x.dtype == torch.float32 and ops.nhwc_ld(x) is None:
                x = ops.images_to_nhwc(x, cpad=cp)
            else:
                x = ops.images_to_nhwc(x.float(), cpad=cp)
        return x, w

    def make_cfg(self, act=L.ACT_NONE, act_param=0.0, bn=None):
        cfg = ops.ConvCfg(self.stride, self.padding, self.dilation, self.groups, act, act_param, has_bn=bn is not None,
                          bn_training=(bn.training or (bn.running_mean is None)) if bn is not None else False,
                          momentum=(bn.momentum if bn is not None and bn.momentum is not None else 0.1),
                          eps=(bn.eps if bn is not None else 1e-5), state=self._hip_state,
                          track=(bn.track_running_stats and bn.training) if bn is not None else False)
        cfg.vkey = (id(self.weight), self.weight._version)
        cfg.sync = sync_of(bn) if bn is not None else None
        cfg.acc_owner = bn   # the layer's persistent statistic accumulators (arena.FlatTrainState) hang on the BatchNorm module
        # flat gradient arena (cvpytorch_amd/arena.py): let backward accumulate straight into the parameters' slots
        ar = getattr(self.weight, "_hip_arena", None)
        if ar is not None and torch.is_grad_enabled():
            cfg.arena = ar[0]
            cfg.gw, cfg.idx_w = self.weight._hip_grad, ar[1]
            if self.bias is not None and getattr(self.bias, "_hip_arena", None) is not None:
                cfg.gb, cfg.idx_b = self.bias._hip_grad, self.bias._hip_arena[1]
            if bn is not None and bn.weight is not None and bn.bias is not None and getattr(bn.weight, "_hip_arena", None) is not None:
                cfg.gg, cfg.gbeta = bn.weight._hip_grad, bn.bias._hip_grad
                cfg.idx_bn = (bn.weight._hip_arena[1], bn.bias._hip_arena[1])
        return cfg

    def forward(self, x):
        x, w = self._effective(x)
        return ops.conv_bn_act(x, w, self.bias, None, None, None, None, None, self.make_cfg())


class HipBN(nn.BatchNorm2d):
    """nn.BatchNorm2d on the HIP engine (NHWC bf16 activations, fp32 statistics)."""

    def forward(self, x):
        training = self.training or self.running_mean is None
        bn_tick(self)
        if self.momentum is None:
            raise L.CvhipError("HipBN: cumulative moving average (momentum=None) is not supported")
        return ops.bn_act(x, self.weight, self.bias, self.running_mean, self.running_var, None, True, training,
                          self.momentum, self.eps, L.ACT_NONE, 0.0, self.track_running_stats and self.training, sync_of(self))


def sync_of(bn):
    """(comm, world) when `bn` is a HipSyncBN whose statistics must be shared right now, else None. The transport is the
    layer's own `comm`, else the process-wide RCCL communicator (comm.set_default), else — test transport — the layer's
    torch.distributed process group."""
    if not isinstance(bn, HipSyncBN) or not (bn.training or bn.running_mean is None):
        return None
    from . import comm as CM
    c = bn.comm if bn.comm is not None else CM.default_comm(bn.process_group)
    if c is None or c.world <= 1:
        return None
    # (the transport is NOT cached on the module: an RcclComm holds a ctypes handle, which would break deepcopy / pickling of the
    # model — ModelEMA, FlatTrainState, checkpoints — after the first training forward; default_comm returns the shared instance)
    return (c, c.world)


class HipSyncBN(HipBN):
    """nn.SyncBatchNorm semantics (trainer.py:126-127) on the HIP engine: batch statistics are the statistics of the GLOBAL
    batch — forward all-reduces the per-channel (sum, sum of squares), backward all-reduces (sum dy, sum dy*xhat); 2K floats
    each, per layer. Equal per-rank batches are assumed (DistributedSampler). Eval mode and single-process runs behave as HipBN.
    Over the RCCL communicator (cvhip_comm_allreduce on the caller's stream) the exchanges are captured into the step's hipGraph
    like any kernel; over the gloo test transport the step runs eagerly."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, process_group=None, comm=None):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.process_group = process_group
        self.comm = comm


def convert_sync_batchnorm(module, process_group=None):
    """torch.nn.SyncBatchNorm.convert_sync_batchnorm for Hip / torch BatchNorm2d layers (parameters and buffers are shared,
    state_dict keys unchanged)."""
    out = module
    if isinstance(module, nn.BatchNorm2d) and not isinstance(module, HipSyncBN):
        out = HipSyncBN(module.num_features, module.eps, module.momentum, module.affine, module.track_running_stats, process_group)
        out.weight, out.bias = module.weight, module.bias
        out.running_mean, out.running_var, out.num_batches_tracked = module.running_mean, module.running_var, module.num_batches_tracked
        out.training = module.training
    for name, child in list(module.named_children()):
        new = convert_sync_batchnorm(child, process_group)
        if new is not child:
            setattr(out, name, new)
    return out


class _HipAct:
    _act = L.ACT_NONE

    def _param(self):
        return 0.0

    def forward(self, x):
        return ops.bn_act(x, has_bn=False, training=False, act=self._act, act_param=self._param())


class HipSiLU(_HipAct, nn.SiLU):
    _act = L.ACT_SILU


class HipReLU(_HipAct, nn.ReLU):
    _act = L.ACT_RELU


class HipLeakyReLU(_HipAct, nn.LeakyReLU):
    _act = L.ACT_LEAKY

    def _param(self):
        return float(self.negative_slope)


class HipSigmoid(_HipAct, nn.Sigmoid):
    _act = L.ACT_SIGMOID


class HipHardswish(_HipAct, nn.Hardswish):
    _act = L.ACT_HSWISH


class Swish(nn.Module):
    """x * sigmoid(x) — src/models/bricks/swish.py:8-25 (stock two-op module; kept for API parity)."""

    def forward(self, x):
        return x * torch.sigmoid(x)


class HipSwish(_HipAct, Swish):
    _act = L.ACT_SILU


class HipMaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        k, s, p = _pair(self.kernel_size), _pair(self.stride if self.stride is not None else self.kernel_size), _pair(self.padding)
        if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or _pair(self.dilation) != (1, 1) or self.ceil_mode or self.return_indices:
            raise L.CvhipError("HipMaxPool2d: only square window, dilation 1, floor mode")
        return ops.max_pool2d(x, k[0], s[0], p[0])


class HipUpsampleNearest2x(nn.Module):
    """nn.UpsamplingNearest2d(scale_factor=2) / nn.Upsample(scale_factor=2, mode='nearest')."""

    def __init__(self, scale_factor=2, mode="nearest", size=None, align_corners=None):
        super().__init__()
        if size is not None or mode != "nearest" or float(scale_factor) != 2.0:
            raise L.CvhipError("HipUpsampleNearest2x: nearest x2 only")
        self.scale_factor = 2

    def forward(self, x):
        return ops.upsample2x_cat(x, None)


class HipAdaptiveAvgPool1x1(nn.Module):
    def forward(self, x):
        return ops.global_avg_pool(x)


class HipAvgPool2d(nn.AvgPool2d):
    """nn.AvgPool2d(k, stride, pad) with count_include_pad=True (torch default) == depthwise conv with constant 1/k^2
    taps, so it runs on libcvhip's depthwise kernels (cvhip_dwconv2d_fprop / _dgrad). Used by STDC's CatBottleneck skip
    (src/models/backbones/seg/stdcnet.py:92)."""

    def forward(self, x):
        k = self.kernel_size if isinstance(self.kernel_size, int) else self.kernel_size[0]
        s = self.stride if isinstance(self.stride, int) else self.stride[0]
        p = self.padding if isinstance(self.padding, int) else self.padding[0]
        if self.ceil_mode or not self.count_include_pad or self.divisor_override is not None:
            raise L.CvhipError("HipAvgPool2d: only floor mode with count_include_pad=True")
        Cc = x.shape[1]
        dw = self.__dict__.get("_dw")
        if dw is None or dw.in_channels != Cc or dw.weight.device != x.device:
            dw = HipConv2d(Cc, Cc, k, s, p, groups=Cc, bias=False).to(x.device)
            dw.weight.data.fill_(1.0 / (k * k))
            dw.weight.requires_grad_(False)
            self.__dict__["_dw"] = dw  # not a registered sub-module: state_dict stays parameter-free like nn.AvgPool2d's
        return dw(x)


# ---- registrations: the reference's stock names (bricks/conv.py:8-9, norm.py:12-16, activation.py:13-30,
# upsample.py:11-12) keep their stock torch classes; the Hip* names select the MI355X engine -------------
CONV_LAYERS.register_module("Conv1d", module=nn.Conv1d)
CONV_LAYERS.register_module("Conv2d", module=nn.Conv2d)
CONV_LAYERS.register_module("HipConv2d", module=HipConv2d)
NORM_LAYERS.register_module("BN", module=nn.BatchNorm2d)
NORM_LAYERS.register_module("BN2d", module=nn.BatchNorm2d)
NORM_LAYERS.register_module("SyncBN", module=nn.SyncBatchNorm)
NORM_LAYERS.register_module("HipSyncBN", module=HipSyncBN)
NORM_LAYERS.register_module("GN", module=nn.GroupNorm)
NORM_LAYERS.register_module("HipBN", module=HipBN)
for _m in (nn.ReLU, nn.LeakyReLU, nn.PReLU, nn.ReLU6, nn.ELU, nn.Sigmoid, nn.Tanh, nn.SiLU, nn.Hardswish):
    ACTIVATION_LAYERS.register_module(module=_m)
ACTIVATION_LAYERS.register_module("Swish", module=Swish)
ACTIVATION_LAYERS.register_module("HipSiLU", module=HipSiLU)
ACTIVATION_LAYERS.register_module("HipSwish", module=HipSwish)
ACTIVATION_LAYERS.register_module("HipReLU", module=HipReLU)
ACTIVATION_LAYERS.register_module("HipLeakyReLU", module=HipLeakyReLU)
ACTIVATION_LAYERS.register_module("HipSigmoid", module=HipSigmoid)
ACTIVATION_LAYERS.register_module("HipHardswish", module=HipHardswish)
UPSAMPLE_LAYERS.register_module("nearest", module=nn.Upsample)
UPSAMPLE_LAYERS.register_module("bilinear", module=nn.Upsample)
UPSAMPLE_LAYERS.register_module("hip_nearest", module=HipUpsampleNearest2x)
PADDING_LAYERS.register_module("zero", module=nn.ZeroPad2d)
PADDING_LAYERS.register_module("reflect", module=nn.ReflectionPad2d)
PADDING_LAYERS.register_module("replicate", module=nn.ReplicationPad2d)


def build_conv_layer(cfg, *args, **kwargs):
    """bricks/conv.py:14-46 — cfg None => 'Conv2d'."""
    if cfg is None:
        cfg_ = dict(type="Conv2d")
    else:
        if not isinstance(cfg, dict):
            raise TypeError("cfg must be a dict")
        if "type" not in cfg:
            raise KeyError('the cfg dict must contain the key "type"')
        cfg_ = cfg.copy()
    layer_type = cfg_.pop("type")
    if layer_type not in CONV_LAYERS:
        raise KeyError("Unrecognized layer type %s" % layer_type)
    return CONV_LAYERS.get(layer_type)(*args, **kwargs, **cfg_)


def infer_abbr(class_type):
    """bricks/norm.py:25-71."""
    if not inspect.isclass(class_type):
        raise TypeError("class_type must be a type, but got %s" % type(class_type))
    if hasattr(class_type, "_abbr_"):
        return class_type._abbr_
    if issubclass(class_type, nn.modules.instancenorm._InstanceNorm):
        return "in"
    if issubclass(class_type, _BatchNorm):
        return "bn"
    if issubclass(class_type, nn.GroupNorm):
        return "gn"
    if issubclass(class_type, nn.LayerNorm):
        return "ln"
    name = class_type.__name__.lower()
    for k in ("batch", "group", "layer", "instance"):
        if k in name:
            return {"batch": "bn", "group": "gn", "layer": "ln", "instance": "in"}[k]
    return "norm_layer"


def build_norm_layer(cfg, num_features, postfix=""):
    """bricks/norm.py:74-123 — returns (name, layer); default eps 1e-5; requires_grad honoured."""
    if not isinstance(cfg, dict):
        raise TypeError("cfg must be a dict")
    if "type" not in cfg:
        raise KeyError('the cfg dict must contain the key "type"')
    cfg_ = cfg.copy()
    layer_type = cfg_.pop("type")
    if layer_type not in NORM_LAYERS:
        raise KeyError("Unrecognized norm type %s" % layer_type)
    norm_layer = NORM_LAYERS.get(layer_type)
    abbr = infer_abbr(norm_layer)
    assert isinstance(postfix, (int, str))
    name = abbr + str(postfix)
    requires_grad = cfg_.pop("requires_grad", True)
    cfg_.setdefault("eps", 1e-5)
    if layer_type != "GN":
        layer = norm_layer(num_features, **cfg_)
    else:
        assert "num_groups" in cfg_
        layer = norm_layer(num_channels=num_features, **cfg_)
    for param in layer.parameters():
        param.requires_grad = requires_grad
    return name, layer


def build_activation_layer(cfg):
    """bricks/activation.py:86-98."""
    return build_from_cfg(cfg, ACTIVATION_LAYERS)


def build_padding_layer(cfg, *args, **kwargs):
    if not isinstance(cfg, dict):
        raise TypeError("cfg must be a dict")
    if "type" not in cfg:
        raise KeyError('the cfg dict must contain the key "type"')
    cfg_ = cfg.copy()
    padding_type = cfg_.pop("type")
    if padding_type not in PADDING_LAYERS:
        raise KeyError("Unrecognized padding type %s." % padding_type)
    return PADDING_LAYERS.get(padding_type)(*args, **kwargs, **cfg_)


def build_upsample_layer(cfg, *args, **kwargs):
    """bricks/upsample.py:53-87."""
    if not isinstance(cfg, dict):
        raise TypeError("cfg must be a dict, but got %s" % type(cfg))
    if "type" not in cfg:
        raise KeyError('the cfg dict must contain the key "type", but got %s' % cfg)
    cfg_ = cfg.copy()
    layer_type = cfg_.pop("type")
    if layer_type not in UPSAMPLE_LAYERS:
        raise KeyError("Unrecognized upsample type %s" % layer_type)
    upsample = UPSAMPLE_LAYERS.get(layer_type)
    if upsample is nn.Upsample:
        cfg_["mode"] = layer_type
    return upsample(*args, **kwargs, **cfg_)


def build_plugin_layer(cfg, postfix="", **kwargs):
    """bricks/plugin.py:58-94 — returns (name, layer)."""
    if not isinstance(cfg, dict):
        raise TypeError("cfg must be a dict")
    if "type" not in cfg:
        raise KeyError('the cfg dict must contain the key "type"')
    cfg_ = cfg.copy()
    layer_type = cfg_.pop("type")
    if layer_type not in PLUGIN_LAYERS:
        raise KeyError("Unrecognized plugin type %s" % layer_type)
    plugin_layer = PLUGIN_LAYERS.get(layer_type)
    abbr = getattr(plugin_layer, "_abbr_", plugin_layer.__name__.lower())
    assert isinstance(postfix, (int, str))
    return abbr + str(postfix), plugin_layer(**kwargs, **cfg_)


HIP_CONV = dict(type="HipConv2d")


def hip_norm(cfg=None):
    c = dict(cfg) if cfg else dict(type="BN")
    if c.get("type") in ("BN", "BN2d"):
        c["type"] = "HipBN"
    elif c.get("type") == "SyncBN":
        c["type"] = "HipSyncBN"
    return c


def hip_act(cfg):
    if cfg is None:
        return None
    c = dict(cfg)
    m = {"SiLU": "HipSiLU", "Swish": "HipSwish", "ReLU": "HipReLU", "LeakyReLU": "HipLeakyReLU", "Sigmoid": "HipSigmoid",
         "Hardswish": "HipHardswish"}
    if c.get("type") in m:
        c["type"] = m[c["type"]]
    if c["type"] in ("HipSwish", "HipSigmoid"):
        c.pop("inplace", None)
    return c


@PLUGIN_LAYERS.register_module(name=["HipConvModule"])
class HipConvModule(nn.Module):
    """ConvModule (src/models/bricks/conv_module.py:20-214) with the same constructor, attributes,
    sub-module names (`conv`, `bn`, `act`) and `forward(x, activate=True, norm=True)`; when the layers
    are Hip layers in ('conv','norm','act') order the whole block runs as ONE fused autograd op:
    MFMA conv with BN partial sums in the epilogue -> finalize -> one BN+act(+residual) pass.

    Default layer types are the Hip ones (conv_cfg None => HipConv2d, 'BN' => HipBN, 'SiLU' => HipSiLU...).
    """

    _abbr_ = "conv_block"

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias="auto",
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True, with_spectral_norm=False,
                 padding_mode="zeros", order=("conv", "norm", "act")):
        super().__init__()
        assert conv_cfg is None or isinstance(conv_cfg, dict)
        assert norm_cfg is None or isinstance(norm_cfg, dict)
        assert act_cfg is None or isinstance(act_cfg, dict)
        if with_spectral_norm or padding_mode != "zeros":
            raise L.CvhipError("HipConvModule: spectral norm / non-zero padding modes are not supported")
        conv_cfg = HIP_CONV if conv_cfg is None or conv_cfg.get("type") == "Conv2d" else conv_cfg
        norm_cfg = hip_norm(norm_cfg) if norm_cfg is not None else None
        act_cfg = hip_act(act_cfg)
        self.conv_cfg, self.norm_cfg, self.act_cfg = conv_cfg, norm_cfg, act_cfg
        self.inplace = inplace
        self.with_spectral_norm = False
        self.with_explicit_padding = False
        self.order = order
        assert isinstance(self.order, tuple) and len(self.order) == 3
        assert set(order) == {"conv", "norm", "act"}
        self.with_norm = norm_cfg is not None
        self.with_act = act_cfg is not None
        if bias == "auto":
            bias = not self.with_norm  # conv_module.py:108-110
        self.with_bias = bias
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                     dilation=dilation, groups=groups, bias=bias)
        self.in_channels = self.conv.in_channels
        self.out_channels = self.conv.out_channels
        self.kernel_size = self.conv.kernel_size
        self.stride = self.conv.stride
        self.padding = padding
        self.dilation = self.conv.dilation
        self.transposed = self.conv.transposed
        self.output_padding = self.conv.output_padding
        self.groups = self.conv.groups
        if self.with_norm:
            norm_channels = out_channels if order.index("norm") > order.index("conv") else in_channels
            self.norm_name, norm = build_norm_layer(norm_cfg, norm_channels)
            self.add_module(self.norm_name, norm)
            if self.with_bias and isinstance(norm, _BatchNorm):
                warnings.warn("Unnecessary conv bias before batch/instance norm")
        else:
            self.norm_name = None
        if self.with_act:
            act_cfg_ = act_cfg.copy()
            if act_cfg_["type"] not in ["Tanh", "PReLU", "Sigmoid", "HSigmoid", "Swish", "GELU", "HipSwish", "HipSigmoid"]:
                act_cfg_.setdefault("inplace", inplace)
            self.act = build_activation_layer(act_cfg_)
        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.norm_name else None

    def init_weights(self):
        """conv_module.py:180-199: kaiming-normal(fan_out) conv, BN weight 1 / bias 0."""
        if not hasattr(self.conv, "init_weights"):
            if self.with_act and self.act_cfg["type"] in ("LeakyReLU", "HipLeakyReLU"):
                nonlinearity, a = "leaky_relu", self.act_cfg.get("negative_slope", 0.01)
            else:
                nonlinearity, a = "relu", 0
            nn.init.kaiming_normal_(self.conv.weight, a=a, mode="fan_out", nonlinearity=nonlinearity)
            if getattr(self.conv, "bias", None) is not None:
                nn.init.constant_(self.conv.bias, 0)
        if self.with_norm:
            if getattr(self.norm, "weight", None) is not None:
                nn.init.constant_(self.norm.weight, 1)
            if getattr(self.norm, "bias", None) is not None:
                nn.init.constant_(self.norm.bias, 0)

    def _fusable(self, activate, norm):
        if self.order != ("conv", "norm", "act") or not isinstance(self.conv, HipConv2d):
            return None
        bn = self.norm if (norm and self.with_norm) else None
        if bn is not None and (not isinstance(bn, HipBN) or bn.momentum is None):
            return None
        act = self.act if (activate and self.with_act) else None
        aid = _act_id(act)
        if aid is None or (act is not None and not isinstance(act, _HipAct)):
            return None
        return bn, aid

    def forward(self, x, activate=True, norm=True, residual=None, out=None, dx_link=None, res_link=None, lazy=False):
        """`out`: optional NHWC channel-slice view that receives the result (concat elimination: the caller hands every
        producer its slice of the concat buffer, ops.cat then has nothing to copy). Ignored on the unfused fallback path.
        `lazy`: the caller passes the result ONLY to Hip conv modules / blocks (which understand ops.LazyAct): in training the
        layer may then skip its BN-apply + activation pass and return its raw convolution output, tagged (ops, round 5)."""
        fus = self._fusable(activate, norm)
        if fus is not None:
            bn, (aid, ap) = fus
            conv = self.conv
            if ops.lazy_of(x) is None:
                x, w = conv._effective(x)
            else:
                w = conv.weight
            cfg = conv.make_cfg(aid, ap, bn)
            cfg.lazy_out = bool(lazy)
            cfg.out = out
            cfg.dx_link, cfg.res_link = dx_link, (res_link if residual is not None else None)
            if bn is not None:
                bn_tick(bn)
                return ops.conv_bn_act(x, w, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual, cfg)
            return ops.conv_bn_act(x, w, conv.bias, None, None, None, None, residual, cfg)
        x = ops.materialize(x)
        if residual is not None:
            residual = ops.materialize(residual)
        for layer in self.order:
            if layer == "conv":
                x = self.conv(x)
            elif layer == "norm" and norm and self.with_norm:
                x = self.norm(x)
            elif layer == "act" and activate and self.with_act:
                x = self.act(x)
        if residual is not None:
            x = ops.add(x, residual)
        return x


ConvModule = HipConvModule  # the name reference blocks import (src/models/bricks/__init__.py)


class HipDepthwiseSeparableConvModule(nn.Module):
    """src/models/bricks/depthwise_separable_conv_module.py:10-99: depthwise ConvModule + pointwise ConvModule."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), dw_norm_cfg="default", dw_act_cfg="default", pw_norm_cfg="default",
                 pw_act_cfg="default", **kwargs):
        super().__init__()
        assert "groups" not in kwargs, "groups should not be specified"
        dw_norm_cfg = dw_norm_cfg if dw_norm_cfg != "default" else norm_cfg
        dw_act_cfg = dw_act_cfg if dw_act_cfg != "default" else act_cfg
        pw_norm_cfg = pw_norm_cfg if pw_norm_cfg != "default" else norm_cfg
        pw_act_cfg = pw_act_cfg if pw_act_cfg != "default" else act_cfg
        self.depthwise_conv = HipConvModule(in_channels, in_channels, kernel_size, stride=stride, padding=padding,
                                            dilation=dilation, groups=in_channels, norm_cfg=dw_norm_cfg, act_cfg=dw_act_cfg, **kwargs)
        self.pointwise_conv = HipConvModule(in_channels, out_channels, 1, norm_cfg=pw_norm_cfg, act_cfg=pw_act_cfg, **kwargs)

    def forward(self, x):
        return self.pointwise_conv(self.depthwise_conv(x))


DepthwiseSeparableConvModule = HipDepthwiseSeparableConvModule


# ------------------------------------------------------------------------------------------------------
# module-swap pass
# ------------------------------------------------------------------------------------------------------
def _swap_conv(m):
    new = HipConv2d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, m.dilation, m.groups,
                    m.bias is not None, m.padding_mode)
    new.weight = m.weight
    new.weight.data = new.weight.data.contiguous(memory_format=torch.channels_last)
    new.bias = m.bias
    return new


def _swap_bn(m):
    if isinstance(m, nn.SyncBatchNorm):
        new = HipSyncBN(m.num_features, m.eps, m.momentum, m.affine, m.track_running_stats, m.process_group)
    else:
        new = HipBN(m.num_features, m.eps, m.momentum, m.affine, m.track_running_stats)
    new.weight, new.bias = m.weight, m.bias
    new.running_mean, new.running_var, new.num_batches_tracked = m.running_mean, m.running_var, m.num_batches_tracked
    new.training = m.training
    return new


def hip_conv_unsupported(m):
    """Why the engine cannot run this nn.Conv2d (None when it can): the conditions HipConv2d's constructor and ops.ConvBnAct refuse"""
    if m.padding_mode != "zeros":
        return "padding_mode=%r (zero padding only)" % (m.padding_mode,)
    if isinstance(m.padding, str):
        return "padding=%r (numeric padding only)" % (m.padding,)
    if m.groups != 1 and not (m.groups == m.in_channels == m.out_channels):
        return "groups=%d with %d -> %d channels (dense or depthwise only)" % (m.groups, m.in_channels, m.out_channels)
    return None


_log = logging.getLogger("cvpytorch_amd")


def convert_to_hip(module):
    """Recursively replace torch layers by their Hip equivalents, sharing Parameters/buffers so
    `state_dict()` keys and values are unchanged (reference checkpoints keep loading:
    src/utils/checkpoints.py:30-41). A layer the engine cannot run (hip_conv_unsupported) is LEFT in place — the idiom of
    SyncBatchNorm.convert_sync_batchnorm (trainer.py:127, src/nn/syncBN.py:10-27), which leaves foreign modules untouched — with one
    logged line; it keeps running on stock torch (NCHW) tensors, so a model that routes engine activations through it must convert
    them itself."""
    out = module
    if type(module) is nn.Conv2d:
        why = hip_conv_unsupported(module)
        if why is None:
            out = _swap_conv(module)
        else:
            _log.warning("convert_to_hip: keeping the stock nn.Conv2d (%d -> %d, k %s): %s", module.in_channels, module.out_channels,
                         tuple(module.kernel_size), why)
    elif type(module) in (nn.BatchNorm2d, nn.SyncBatchNorm):
        out = _swap_bn(module)
    elif type(module) is nn.SiLU:
        out = HipSiLU()
    elif type(module) is nn.ReLU:
        out = HipReLU()
    elif type(module) is nn.LeakyReLU:
        out = HipLeakyReLU(module.negative_slope)
    elif type(module) is Swish:
        out = HipSwish()
    elif type(module) is nn.MaxPool2d:
        out = HipMaxPool2d(module.kernel_size, module.stride, module.padding, module.dilation, module.return_indices, module.ceil_mode)
    elif type(module) is nn.UpsamplingNearest2d or (type(module) is nn.Upsample and module.mode == "nearest"):
        if module.size is None and float(module.scale_factor if not isinstance(module.scale_factor, tuple) else module.scale_factor[0]) == 2.0:
            out = HipUpsampleNearest2x()
    elif type(module) is nn.AdaptiveAvgPool2d and module.output_size in (1, (1, 1)):
        out = HipAdaptiveAvgPool1x1()
    for name, child in module.named_children():
        new_child = convert_to_hip(child)
        if new_child is not child:
            out.add_module(name, new_child)
    return out


class HipConvBN(nn.Sequential):
    """`nn.Sequential(Conv2d(bias=False), BatchNorm2d)` (state_dict keys `0.*`, `1.*`) executed as ONE fused op — the shape
    the reference uses for RepConv branches (yolov7_modules.py:187-195) and STDC's avd/skip layers (stdcnet.py:37-48)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, groups=1, act=L.ACT_NONE):
        super().__init__(HipConv2d(in_channels, out_channels, kernel_size, stride, padding, groups=groups, bias=False),
                         HipBN(out_channels))
        self._act = act

    def forward(self, x, residual=None):
        conv, bn = self[0], self[1]
        bn_tick(bn)
        xx, w = conv._effective(x)
        return ops.conv_bn_act(xx, w, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual,
                               conv.make_cfg(self._act, 0.0, bn))
