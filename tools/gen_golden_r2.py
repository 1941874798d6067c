"""Round-2 golden vectors, same method as tools/gen_golden.py (the reference's own classes imported in THIS container, seeded
inputs, small .npz fixtures): the generic CSPDarknet backbone (src/models/backbones/det/csp_darknet.py:25-103), plain and
depthwise.

    python tools/gen_golden_r2.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import install, run_module, save  # noqa: E402


def main():
    install()
    from src.models.backbones.det.csp_darknet import CSPDarknet
    for name, kw in (("cspdarknet_n", dict(subtype="cspdark_n")), ("cspdarknet_n_dw", dict(subtype="cspdark_n", depthwise=True))):
        torch.manual_seed(31)
        bb = CSPDarknet(out_stages=[2, 3, 4], **kw)
        for m in bb.modules():   # non-trivial BN state so eval/train statistics matter
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.2)
        x = torch.randn(2, 3, 64, 96)
        state0 = {k: v.clone() for k, v in bb.state_dict().items()}
        outs, cots, gin, gpar = run_module(bb, [x])
        save(name, x=x, state=state0, out=outs, cot=cots, gparam_norms={k: v.norm() for k, v in gpar.items()},
             g_stem=gpar["stem.conv.conv.weight"], out_channels=torch.tensor(bb.out_channels))


if __name__ == "__main__":
    main()
