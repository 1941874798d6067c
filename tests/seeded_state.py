"""Deterministic module state for fixtures that would otherwise have to carry a whole ResNet-50 state_dict (100 MB): every
floating-point tensor of `module.state_dict()` is overwritten from a CPU generator seeded by (seed, crc32 of the tensor's KEY), so
two modules with the same key names / shapes — the reference's class in the build container, the oracle's or the engine's on the
GPU box — end up with identical parameters and buffers whatever their construction order. Test infrastructure only."""
import zlib

import torch


def seed_state(module, seed):
    g = torch.Generator()
    sig = []
    with torch.no_grad():
        for k, v in module.state_dict().items():
            sig.append("%s:%s" % (k, "x".join(str(int(d)) for d in v.shape)))
            if not v.dtype.is_floating_point:
                continue
            g.manual_seed((int(seed) * 1000003 + zlib.crc32(k.encode())) & 0x7fffffff)
            if k.endswith("running_var"):
                t = torch.rand(v.shape, generator=g) + 0.5
            elif k.endswith("running_mean"):
                t = torch.randn(v.shape, generator=g) * 0.1
            elif v.dim() == 1 and k.endswith(".weight"):       # BatchNorm gamma
                t = torch.rand(v.shape, generator=g) * 0.5 + 0.75
            elif v.dim() == 1:                                   # biases / BatchNorm beta
                t = torch.randn(v.shape, generator=g) * 0.1
            else:                                                # conv / linear weights: He-style scale keeps activations O(1)
                fan_in = max(1, int(v[0].numel()))
                t = torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5
            v.copy_(t.to(v.dtype))
    return sig
