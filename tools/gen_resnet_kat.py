"""Hand-derived known-answer file for torchvision's ResNet-50 (tests/golden/resnet50_kat.json).

torchvision is a third-party dependency of the reference (`from torchvision.models.resnet import resnet50`,
src/models/backbones/seg/resnet.py:11,52-54) that is neither vendored nor installed here, so the oracle's Bottleneck / ResNet-50
restatement (oracle/torch_ref.py) cannot be pinned against the binary. What CAN be pinned is everything the published definition fixes
(He et al. 2015, Table 1 + torchvision's documented "v1.5" variant: the stride sits on the 3x3 convolution of a bottleneck):

  * block counts [3, 4, 6, 3], planes [64, 128, 256, 512], expansion 4, a 1x1-conv + BN projection shortcut on the first block of
    every layer, stride 2 in layers 2-4 (on conv2 and on the projection), no conv biases;
  * the PUBLISHED parameter counts: 25,557,032 with the 1000-way classifier, 23,508,032 without it;
  * the state_dict key list (conv1, bn1, layerL.B.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.0,downsample.1}, fc).

The numbers below are derived BY HAND from those rules (the arithmetic is spelled out so that a reader can follow it) and must
reproduce the two published totals — that is the cross-check of the derivation itself. No torch import, no reference import."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BLOCKS, PLANES, EXPANSION = [3, 4, 6, 3], [64, 128, 256, 512], 4


def bottleneck_params(inplanes, planes, projection):
    out = planes * EXPANSION
    n = inplanes * planes            # conv1 1x1 (no bias)
    n += 2 * planes                  # bn1 weight + bias
    n += planes * planes * 9         # conv2 3x3
    n += 2 * planes                  # bn2
    n += planes * out                # conv3 1x1
    n += 2 * out                     # bn3
    if projection:
        n += inplanes * out + 2 * out  # downsample.0 (1x1 conv) + downsample.1 (BN)
    return n


def derive():
    keys, shapes, layer_params, strides = [], {}, {}, {}

    def add(name, shape):
        keys.append(name)
        shapes[name] = list(shape)

    def add_bn(prefix, c):
        add(prefix + ".weight", (c,))
        add(prefix + ".bias", (c,))
        add(prefix + ".running_mean", (c,))
        add(prefix + ".running_var", (c,))
        add(prefix + ".num_batches_tracked", ())

    add("conv1.weight", (64, 3, 7, 7))
    add_bn("bn1", 64)
    stem = 64 * 3 * 49 + 2 * 64      # 9,408 + 128 = 9,536
    inplanes = 64
    total = stem
    for li, (nb, planes) in enumerate(zip(BLOCKS, PLANES), start=1):
        lp = 0
        for b in range(nb):
            pre = "layer%d.%d" % (li, b)
            stride = 2 if (li > 1 and b == 0) else 1
            proj = b == 0            # layer1.0: 64 -> 256 channels; layers 2-4: stride 2 and a channel change
            add(pre + ".conv1.weight", (planes, inplanes, 1, 1))
            add_bn(pre + ".bn1", planes)
            add(pre + ".conv2.weight", (planes, planes, 3, 3))
            add_bn(pre + ".bn2", planes)
            add(pre + ".conv3.weight", (planes * 4, planes, 1, 1))
            add_bn(pre + ".bn3", planes * 4)
            strides[pre + ".conv1"] = 1
            strides[pre + ".conv2"] = stride     # v1.5: the stride is on the 3x3
            strides[pre + ".conv3"] = 1
            if proj:
                add(pre + ".downsample.0.weight", (planes * 4, inplanes, 1, 1))
                add_bn(pre + ".downsample.1", planes * 4)
                strides[pre + ".downsample.0"] = stride
            lp += bottleneck_params(inplanes, planes, proj)
            inplanes = planes * 4
        layer_params["layer%d" % li] = lp
        total += lp
    add("fc.weight", (1000, 2048))
    add("fc.bias", (1000,))
    fc = 2048 * 1000 + 1000
    return {"blocks": BLOCKS, "planes": PLANES, "expansion": EXPANSION, "stem_params": stem, "layer_params": layer_params,
            "fc_params_1000": fc, "total_without_fc": total, "total_with_fc_1000": total + fc, "state_dict_keys": keys, "shapes": shapes,
            "conv_strides": strides,
            "provenance": "hand-derived from the published ResNet-50 definition (He et al. 2015 Table 1; torchvision v1.5 stride placement); "
                          "cross-checked against the published parameter counts 25,557,032 / 23,508,032 in this script"}


if __name__ == "__main__":
    k = derive()
    # worked example, layer1: block 0 = 64*64 + 128 + 64*64*9 + 128 + 64*256 + 512 + (64*256 + 512) = 75,008; blocks 1-2 =
    # 256*64 + 128 + 36,864 + 128 + 16,384 + 512 = 70,400 each => 75,008 + 2*70,400 = 215,808
    assert k["layer_params"] == {"layer1": 215808, "layer2": 1219584, "layer3": 7098368, "layer4": 14964736}, k["layer_params"]
    assert k["total_with_fc_1000"] == 25557032 and k["total_without_fc"] == 23508032   # the PUBLISHED figures
    assert len(k["state_dict_keys"]) == 320
    out = os.path.join(ROOT, "tests", "golden", "resnet50_kat.json")
    json.dump(k, open(out, "w"), indent=0, sort_keys=True)
    print("wrote", out, k["total_with_fc_1000"], k["total_without_fc"])
