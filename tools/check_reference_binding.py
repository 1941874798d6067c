"""Executes INTEGRATION.md §1 / §2 LITERALLY against the reference's own code (build container only; never runs on the GPU box).

  §1  registers cvpytorch_amd's Hip layers into the REFERENCE's registries (/root/reference/src/models/bricks/registry.py:4-9) and
      lets the reference's own builders construct them: `ConvModule` (bricks/conv_module.py:119-168), `CSPLayer`, `SPPF`
      (modules/yolo_modules.py:107-195) and the whole `YOLOv5CSPDarknet` (backbones/det/yolov5_csp_darknet.py:16-91) with
      conv_cfg=dict(type='HipConv2d'), norm_cfg=dict(type='HipBN', momentum=0.03, eps=0.001), act_cfg=dict(type='HipSiLU').
  §2  runs `convert_to_hip` on reference-built `YOLOv5Detect`, `STDCNet` and `Deeplabv3PlusHead`.

For every model it records the module tree (qualified name -> class, with the Hip classes named) and the state_dict keys + shapes
into tests/golden/binding_*.json, after checking on the spot that (a) every conv / norm / activation leaf the reference built IS a
Hip layer, (b) the state_dict keys and shapes equal those of the same reference model built with its stock cfg (checkpoints keep
loading: utils/checkpoints.py:30-41), (c) a CPU forward of the stock model and the Hip-built model cannot be compared here (the Hip
layers have no CPU path by design) — parity of the arithmetic is what tests/test_gpu_modules.py pins with reference fixtures.

tests/test_reference_binding.py (CPU) then asserts that the engine's OWN assembled models (cvpytorch_amd.yolov5 / stdc / deeplab ...)
have exactly these trees and keys. The fixtures are data (names, class names, shapes); no reference source is copied.

    python tools/check_reference_binding.py        # writes tests/golden/binding_*.json
"""
import json
import os
import sys

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gen_golden  # the import recipe (stubs for absent third-party roots)

OUT = os.path.join(ROOT, "tests", "golden")


def tree(model):
    return {name or "<root>": type(m).__name__ for name, m in model.named_modules()}


def keys(model):
    return {k: list(v.shape) for k, v in model.state_dict().items()}


def leaves_are_hip(model, hip):
    """every Conv2d / BatchNorm2d / activation leaf is an instance of the Hip class that replaces it"""
    bad = []
    for name, m in model.named_modules():
        if isinstance(m, nn.Conv2d) and not isinstance(m, hip.HipConv2d):
            bad.append((name, type(m).__name__))
        if isinstance(m, (nn.BatchNorm2d, nn.SyncBatchNorm)) and not isinstance(m, hip.HipBN):
            bad.append((name, type(m).__name__))
        if isinstance(m, (nn.SiLU, nn.ReLU, nn.LeakyReLU)) and not isinstance(m, hip._HipAct):
            bad.append((name, type(m).__name__))
    return bad


def main():
    gen_golden.install()
    from cvpytorch_amd import bricks as hip
    # ---- INTEGRATION.md §1, verbatim ------------------------------------------------------------------------------------
    from src.models.bricks.registry import ACTIVATION_LAYERS, CONV_LAYERS, NORM_LAYERS, PLUGIN_LAYERS, UPSAMPLE_LAYERS
    CONV_LAYERS.register_module('HipConv2d', module=hip.HipConv2d)
    NORM_LAYERS.register_module('HipBN', module=hip.HipBN)
    ACTIVATION_LAYERS.register_module('HipSiLU', module=hip.HipSiLU)
    ACTIVATION_LAYERS.register_module('HipSwish', module=hip.HipSwish)
    ACTIVATION_LAYERS.register_module('HipReLU', module=hip.HipReLU)
    UPSAMPLE_LAYERS.register_module('hip_nearest', module=hip.HipUpsampleNearest2x)
    PLUGIN_LAYERS.register_module('HipConvModule', module=hip.HipConvModule)
    HIPCFG = dict(conv_cfg=dict(type='HipConv2d'), norm_cfg=dict(type='HipBN', momentum=0.03, eps=0.001), act_cfg=dict(type='HipSiLU'))
    STOCK = dict(conv_cfg=None, norm_cfg=dict(type='BN', momentum=0.03, eps=0.001), act_cfg=dict(type='SiLU', inplace=True))

    from src.models.bricks import ConvModule
    from src.models.modules.yolo_modules import CSPLayer, SPPF
    from src.models.backbones.det.yolov5_csp_darknet import YOLOv5CSPDarknet

    report = {}

    def record(name, build):
        m_hip, m_ref = build(HIPCFG), build(STOCK)
        bad = leaves_are_hip(m_hip, hip)
        threaded = True
        if bad:
            # the reference does not thread conv_cfg everywhere (YOLOv5CSPDarknet.build_stem_layer / build_stage_layer pass only
            # norm_cfg / act_cfg to ConvModule: yolov5_csp_darknet.py:38-61): norm and activation layers came out of the registries as
            # Hip layers, the convolutions as nn.Conv2d. INTEGRATION.md §1 + §2: one convert_to_hip() call swaps what is left.
            assert all(cls == "Conv2d" for _, cls in bad), (name, bad[:5])
            threaded = False
            m_hip = hip.convert_to_hip(m_hip)
            bad = leaves_are_hip(m_hip, hip)
        assert not bad, (name, bad[:5])
        assert keys(m_hip) == keys(m_ref), name                     # same checkpoint layout as the stock build
        stock_tree, hip_tree = tree(m_ref), tree(m_hip)
        assert stock_tree.keys() == hip_tree.keys(), name
        # the registries built what the cfg asked for: BN momentum / eps reached the Hip layer
        for mod in m_hip.modules():
            if isinstance(mod, hip.HipBN):
                assert abs(mod.momentum - 0.03) < 1e-12 and abs(mod.eps - 0.001) < 1e-12
        report[name] = dict(tree=hip_tree, stock_tree=stock_tree, state_dict=keys(m_hip), conv_cfg_threaded=threaded)
        print("%-28s %4d modules, %4d state_dict entries, %d Hip conv / %d Hip BN / %d Hip act leaves" % (
            name, len(hip_tree), len(report[name]["state_dict"]), sum(isinstance(x, hip.HipConv2d) for x in m_hip.modules()),
            sum(isinstance(x, hip.HipBN) for x in m_hip.modules()), sum(isinstance(x, hip._HipAct) for x in m_hip.modules())) +
              ("" if threaded else "   [conv_cfg not threaded by the reference: convert_to_hip() swapped the convolutions]"))

    record("convmodule_3x3", lambda c: ConvModule(32, 64, 3, stride=2, padding=1, **c))
    record("csplayer_64_n2", lambda c: CSPLayer(64, 64, n=2, shortcut=True, **c))
    record("sppf_256", lambda c: SPPF(256, 256, kernel_sizes=5, **c))
    record("yolov5_cspdarknet_s", lambda c: YOLOv5CSPDarknet(subtype='cspdark_s', **c))
    record("yolov5_cspdarknet_n", lambda c: YOLOv5CSPDarknet(subtype='cspdark_n', **c))

    # ---- convert_to_hip on models the reference builds with plain torch layers -------------------------
    from src.models.detects.yolov5_detect import YOLOv5Detect
    from src.models.backbones.seg.stdcnet import STDCNet
    from src.models.heads.seg.deeplabv3plus_head import Deeplabv3PlusHead

    def record_swap(name, model):
        before_keys, before_tree = keys(model), tree(model)
        swapped = hip.convert_to_hip(model)
        bad = leaves_are_hip(swapped, hip)
        assert not bad, (name, bad[:5])
        assert keys(swapped) == before_keys, name
        assert tree(swapped).keys() == before_tree.keys(), name
        report[name] = dict(tree=tree(swapped), stock_tree=before_tree, state_dict=keys(swapped))
        print("%-28s %4d modules, %4d state_dict entries (convert_to_hip)" % (name, len(before_tree), len(before_keys)))

    # (num_layers, num_anchors, 2), as models/yolov5.py hands them to the detect layer
    anchors = [[[10, 13], [16, 30], [33, 23]], [[30, 61], [62, 45], [59, 119]], [[116, 90], [156, 198], [373, 326]]]
    record_swap("yolov5_detect_s", YOLOv5Detect(num_classes=80, in_channels=[256, 512, 1024], anchors=anchors, depth_mul=0.33, width_mul=0.5))
    record_swap("stdcnet_stdc1", STDCNet(subtype='stdc1'))
    head = Deeplabv3PlusHead(low_in_channels=256, low_channels=48, num_classes=19, in_channels=2048, channels=512, dilations=(1, 12, 24, 36))
    record_swap("deeplabv3plus_head", head)

    os.makedirs(OUT, exist_ok=True)
    for name, rec in report.items():
        with open(os.path.join(OUT, "binding_%s.json" % name), "w") as fh:
            json.dump(rec, fh, indent=0, sort_keys=True)
    print("wrote %d fixtures to %s" % (len(report), OUT))


if __name__ == "__main__":
    main()
