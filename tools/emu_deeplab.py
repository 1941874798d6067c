"""Dev tool: DeepLabv3+ at 512x1024, batch 2: per-parameter gradient cosine of the engine and of the generic storage emulator
against the fp32 oracle, in network order."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import storage_emulator as E
from oracle import torch_ref as R
from cvpytorch_amd import deeplab
dev = torch.device("cuda:0")
size = tuple(int(v) for v in os.environ.get("SIZE", "512,1024").split(","))
B = int(os.environ.get("B", "2"))
def cos(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm()).clamp(min=1e-30))
torch.manual_seed(0)
ref = R.EncoderDecoder(19, output_stride=32, dropout_ratio=0).train()
imgs, tgt = R.synthetic_seg_batch(B, size, seed=3)
lr = ref(imgs, tgt, "train")["loss"]; lr.backward()
emu = R.EncoderDecoder(19, output_stride=32, dropout_ratio=0).train(); emu.load_state_dict(ref.state_dict())
E.emulate_storage_generic(emu, torch.bfloat16)
le = emu(imgs, tgt, "train")["loss"]; le.backward()
hip = deeplab.EncoderDecoder(19, output_stride=32, dropout_ratio=0); hip.load_state_dict(ref.state_dict()); hip.to(dev).train()
lh = hip(imgs.to(dev), tgt.to(dev), "train")["loss"]; lh.backward(); torch.cuda.synchronize()
print("loss oracle %.5f emulator %.5f engine %.5f" % (float(lr), float(le), float(lh)))
rp, ep = dict(ref.named_parameters()), dict(emu.named_parameters())
rows = [(n, cos(p.grad.float(), rp[n].grad), cos(ep[n].grad, rp[n].grad), cos(p.grad.float(), ep[n].grad), float(p.grad.float().norm()) / max(float(rp[n].grad.norm()), 1e-30)) for n, p in hip.named_parameters()]
print("median engine~oracle %.4f  emulator~oracle %.4f  engine~emulator %.4f" % tuple(np.median([r[i] for r in rows]) for i in (1, 2, 3)))
for r in rows[::6]:
    print("  %-50s engine~oracle %7.4f  emu~oracle %7.4f  engine~emu %7.4f  |g| ratio %.3f" % r)
