"""Host-side logic of the Classification wrapper: the fixed-shape per-class CE terms equal the
reference's boolean-mask formulation (restated in oracle/cls_ref.py). Pure torch on logits — no GPU needed."""
import torch
import torch.nn.functional as F

from cvpytorch_amd import classification as C
from oracle import cls_ref as RC


def test_per_class_terms_equal_masked_reference_form():
    torch.manual_seed(0)
    dictionary = [{"c%d" % i: 0.5 + 0.1 * i} for i in range(7)]
    logits = torch.randn(16, 7)
    targets = torch.randint(0, 7, (16,))
    targets[3] = 5

    class _Stub(torch.nn.Module):
        def forward(self, x):
            return logits

    m = C.Classification(dictionary)
    m.backbone = _Stub()
    got = m(None, targets, "train")
    r = RC.Classification(dictionary)
    r.backbone = _Stub()
    exp = r(None, targets, "train")
    assert set(got) == set(exp)
    for k in exp:
        assert torch.allclose(got[k], exp[k], rtol=1e-6, atol=1e-7), k
    lv, preds = m(None, targets, "val")
    assert torch.equal(preds, logits.argmax(1))
    assert torch.allclose(m(None, None, "infer"), F.softmax(logits, 1))


def test_state_dict_keys_match_oracle():
    a = C.Classification(num_classes=10).state_dict()
    b = RC.Classification([{"c%d" % i: 1.0} for i in range(10)]).state_dict()
    assert set(a) == set(b) - {"criterion.weight"}
