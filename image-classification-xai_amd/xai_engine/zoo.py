"""Classifier architectures for synthetic benchmarks and tests (seeded random weights -- there
is no network for checkpoints).  The hot path is model-agnostic: these exist only so that
bench.py can run BASELINE.json's named configurations (ResNet-50, ViT-B/16).

ResNet-50 is the standard v1.5 bottleneck network (stride on the 3x3 conv), the architecture
behind torchvision.models.resnet50; resnet101 / resnet152 / resnext101_64x4d are the deeper siblings the
reference harness instantiates (evaluatePerturbation.py:627-647: "R101", "R152", "RNXT"), same block with more
layers / grouped 3x3 convolutions, parameter names as in torchvision so that its state dicts load as they are.  ViT-B/16 follows the
layout of the reference's hooked model (util/attribution_methods/VIT_LRP/ViT_ig.py:57-253:
patch 16, dim 768, depth 12, 12 heads, qkv bias, LayerNorm eps 1e-6, class token, learned
position embedding) including its attention-map / attention-gradient hooks.
"""
import torch
import torch.nn as nn


# ------------------------------------------------------------------------------ ResNet-50
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups          # ResNeXt: wider, grouped 3x3 (torchvision's rule)
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class ResNet(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), num_classes=1000, width=64, groups=1, width_per_group=64):
        super().__init__()
        self.groups, self.base_width = groups, width_per_group
        self.inplanes = width
        self.conv1 = nn.Conv2d(3, width, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make(width, layers[0], 1)
        self.layer2 = self._make(width * 2, layers[1], 2)
        self.layer3 = self._make(width * 4, layers[2], 2)
        self.layer4 = self._make(width * 8, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(width * 8 * 4, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        seq = [Bottleneck(self.inplanes, planes, stride, down, self.groups, self.base_width)]
        self.inplanes = planes * 4
        seq += [Bottleneck(self.inplanes, planes, groups=self.groups, base_width=self.base_width) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def _frozen_resnet(seed, layers, num_classes, width, groups=1, width_per_group=64):
    torch.manual_seed(seed)
    m = ResNet(layers, num_classes, width, groups, width_per_group).eval()
    for p in m.parameters():
        p.requires_grad_(False)
    return m


def resnet50(seed=0, num_classes=1000, width=64):
    return _frozen_resnet(seed, (3, 4, 6, 3), num_classes, width)


def resnet101(seed=0, num_classes=1000, width=64):
    """the reference's "R101" (evaluatePerturbation.py:627-633)"""
    return _frozen_resnet(seed, (3, 4, 23, 3), num_classes, width)


def resnet152(seed=0, num_classes=1000, width=64):
    """the reference's "R152" (:634-640; its table builds resnet101 and asks for ResNet152 weights -- here the architecture
    the weights belong to)"""
    return _frozen_resnet(seed, (3, 8, 36, 3), num_classes, width)


def resnext101_64x4d(seed=0, num_classes=1000, width=64):
    """the reference's "RNXT" (:641-647): 64 groups, 4 channels per group"""
    return _frozen_resnet(seed, (3, 4, 23, 3), num_classes, width, groups=64, width_per_group=4)


# ------------------------------------------------------------------------------ ViT-B/16 with hooks
class Attention(nn.Module):
    def __init__(self, dim, heads, qkv_bias=True):
        super().__init__()
        self.heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.attention_map = None
        self.attn_gradients = None
        self.output = None

    def save_attn_gradients(self, g):
        self.attn_gradients = g

    def get_output(self):
        """projected attention output of the last forward pass (reference ViT_new_timm.py:223-227,251): read by the InFlow rollout"""
        return self.output

    def get_attn_gradients(self):
        return self.attn_gradients

    def get_attention_map(self):
        return self.attention_map

    def forward(self, x, register_hook=False):
        B, N, D = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.heads, D // self.heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
        self.attention_map = attn
        if register_hook and attn.requires_grad:
            attn.register_hook(self.save_attn_gradients)
        self.output = self.proj((attn @ v).transpose(1, 2).reshape(B, N, D))
        return self.output


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

        self.block_out = None
        self.input = self.input_plus_attn = self.mlp_val = None

    def get_block_out(self):
        """output of the last forward pass (reference ViT_new_timm.py:285,297-312): what `single_run(return_embeddings=True)` reads"""
        return self.block_out

    # the residual stream of the last forward pass (reference ViT_new_timm.py:276-283,297-312): what the InFlow rollout weighs
    def get_input(self):
        return self.input

    def get_input_plus_attn(self):
        return self.input_plus_attn

    def get_mlp_val(self):
        return self.mlp_val

    def forward(self, x, register_hook=False):
        self.input = x
        x = x + self.attn(self.norm1(x), register_hook)
        self.input_plus_attn = x
        y = self.mlp(self.norm2(x))
        self.mlp_val = y
        x = x + y
        self.block_out = x
        return x


class PatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.patch_size = (patch, patch)
        self.proj = nn.Conv2d(3, dim, patch, stride=patch)

    def forward(self, x):
        """Non-overlapping patches: the stride-p p x p convolution IS a GEMM over (C*p*p)-vectors, and is run as one --
        same weights (`proj.weight` / `proj.bias`, so state dicts interchange with the reference's Conv2d), same sums.
        MIOpen has no tuned solver for this shape: in immediate mode its backward-data (the gradient with respect to the
        image, which every IG step needs) falls to a naive kernel at ~7 s per call."""
        p = self.patch_size[0]
        B, C, H, W = x.shape
        if H % p == 0 and W % p == 0:
            patches = x.reshape(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // p) * (W // p), C * p * p)
            return nn.functional.linear(patches, self.proj.weight.reshape(self.proj.weight.shape[0], -1), self.proj.bias)
        return self.proj(x).flatten(2).transpose(1, 2)


class VisionTransformer(nn.Module):
    """Parameter names follow the reference's ViT_ig.py / timm (patch_embed.proj, blocks.N.attn.qkv,
    blocks.N.mlp.fc1 ...), so state dicts interchange."""

    def __init__(self, img=224, patch=16, dim=768, depth=12, heads=12, num_classes=1000):
        super().__init__()
        self.patch_embed = PatchEmbed(patch, dim)
        n = (img // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, dim))
        self.pos_drop = nn.Identity()        # the reference's Dropout(p=0) (ViT_ig.py:175): TIS hangs its token-sampling hook here
        self.blocks = nn.ModuleList([Block(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, num_classes)
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x, register_hook=False):
        B = x.shape[0]
        x = self.patch_embed(x)
        x = self.pos_drop(torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1) + self.pos_embed)
        for blk in self.blocks:
            x = blk(x, register_hook)
        return self.head(self.norm(x)[:, 0])


def vit_base_patch16_224(seed=0, **kw):
    torch.manual_seed(seed)
    m = VisionTransformer(**kw).eval()
    for p in m.parameters():
        p.requires_grad_(False)
    return m


def vit_base_patch32_224(seed=0, **kw):
    """the reference's "VIT32" (evaluatePerturbation.py:654-659): patch 32, 7 x 7 patches"""
    return vit_base_patch16_224(seed, patch=32, **kw)
