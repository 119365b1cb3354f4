"""Attention-space attributions for hooked ViTs (reference
util/attribution_methods/VIT_LRP/ViT_explanation_generator.py: Baselines :135-386).

The model contract is the reference's: `model(x, register_hook=True)` records the last block's
attention map and, on backward, its gradient (`blocks[-1].attn.get_attention_map()` /
`.get_attn_gradients()`, ViT_ig.py:73-111).

`Baselines.IG` is a K2-shaped accumulation: the reference runs `steps` sequential
forward/backward passes on `input * alpha` and sums the (1,heads,S,S) attention gradients, of which
only the CLS row survives (`.mean(1)[:, 0, :]`, :381).  Here the `steps` scaled inputs go through
the classifier as ONE batch, the CLS rows (steps, heads, S) are reduced over the step axis by
xai_ig_accum_f32, and the clamp / head-mean finish on 2 364 numbers.
"""
import numpy as np
import torch

from . import kernels as K
from .ig import hip_device
from .streams import backward_turn


def _backward(score):
    """score.backward() as one turn of the device's backward passes (xai_engine/streams.py)"""
    if score.is_cuda:
        with backward_turn(score.device):
            score.backward()
    else:
        score.backward()


class Baselines:
    def __init__(self, model):
        self.model = model
        self.model.eval()

    def _last_attn(self):
        return self.model.blocks[-1].attn

    def generate_raw_attn(self, input, device, layer=-1):
        """Head-mean CLS attention of one block (reference :139-144)."""
        with torch.no_grad():
            self.model(input.to(device))
        attn = self.model.blocks[layer].attn.get_attention_map().mean(1)[0, 0, 1:]
        side = int(np.sqrt(attn.shape[-1]))
        return attn.reshape(-1, side, side)

    def generate_grad(self, input, target_class, device, layer=-1):
        """Clamped head-mean CLS attention gradient (reference :146-157)."""
        x = input.to(device).detach().requires_grad_(True)
        output = self.model(x, register_hook=True)
        _backward(output[0][target_class].sum())
        grad = self.model.blocks[layer].attn.get_attn_gradients().mean(1)[:, 0, 1:].clamp(0)
        side = int(np.sqrt(grad.shape[-1]))
        return grad.reshape(-1, side, side)

    def IG(self, input, target_class, steps=20, device="cuda:0"):
        """Attention-space Integrated Gradients (reference :358-386) -> (1, side, side)."""
        dev = hip_device(device)
        x = input.to(dev, torch.float32)
        alphas = torch.from_numpy(np.linspace(0, 1, steps)).to(dev, torch.float32)    # float64 linspace rounded once
        scaled = (x * alphas.reshape(steps, 1, 1, 1)).detach().requires_grad_(True)   # (steps,C,H,W): input * alpha
        output = self.model(scaled, register_hook=True)
        _backward(output[:, target_class].sum())
        g = self._last_attn().get_attn_gradients()                                    # (steps, heads, S, S)
        heads, S = g.shape[1], g.shape[-1]
        cls_rows = g[:, :, 0, :].contiguous().reshape(1, steps, heads, S)              # the only rows that are used
        ones = torch.ones((1, heads, S), dtype=torch.float32, device=dev)
        mean = K.ig_accum(cls_rows, ones, 0.0)[0]                                      # sum_s / steps, (heads, S)
        w = mean.clamp(min=0).mean(0)                                                  # (S,)
        side = int(np.sqrt(S - 1))
        return w[1:].reshape(-1, side, side)


def compute_rollout_naive(all_layer_matrices, start_layer=0):
    """Product of the head-averaged attention matrices, no residual (reference :14-24)."""
    mats = torch.stack(all_layer_matrices)
    joint = mats[start_layer]
    for i in range(start_layer + 1, mats.shape[0]):
        joint = mats[i].bmm(joint)
    return joint, mats


def compute_rollout_attention(all_layer_matrices, start_layer=0):
    """Attention rollout with the residual modelled as 0.5*A + 0.5*I (reference :26-45)."""
    n_tok, bsz = all_layer_matrices[0].shape[1], all_layer_matrices[0].shape[0]
    eye = torch.eye(n_tok, device=all_layer_matrices[0].device).expand(bsz, n_tok, n_tok)
    aug = torch.stack(all_layer_matrices) + eye.unsqueeze(0)
    aug = aug / aug.sum(dim=-1, keepdim=True)
    joint = aug[start_layer]
    for i in range(start_layer + 1, aug.shape[0]):
        joint = aug[i].bmm(joint)
    return joint, aug


def _residual_shares(blk):
    """Per token, how the two branches of a block's two residual additions share the 2-norm (reference :214-235, :452-460):
    ((input, attention output), (input + attention, MLP output)), each a (2, S) tensor whose columns sum to 1.  The reference
    `.squeeze()`s the batch axis away: one image."""
    def share(a, b):
        both = torch.stack((torch.linalg.norm(a.squeeze(0), ord=2, dim=1), torch.linalg.norm(b.squeeze(0), ord=2, dim=1)))
        return torch.nn.functional.normalize(both, p=1, dim=0)
    return share(blk.get_input().detach(), blk.attn.get_output().detach()), share(blk.get_input_plus_attn().detach(), blk.get_mlp_val().detach())


def compute_RAVE(all_layer_attentions, all_layer_biases_resid_1, all_layer_biases_resid_2, ablate=0):
    """InFlow rollout (reference compute_RAVE :48-88): each block's head-mean attention A is mixed with the identity by the norm
    shares of its first residual addition (A * attn_share + I * input_share), multiplied by the second one's matrix
    (diag(normalised mlp/resid ratio) * diag(mlp_share) + I * diag(resid_share)), row-normalised, and the blocks are chained.
    Tensors of (L, 1, S, S) / (L, 2, S) for one image -- classifier-side bookkeeping, device torch ops."""
    A = torch.stack(all_layer_attentions)                                          # (L, B, S, S)
    b1 = torch.stack(all_layer_biases_resid_1).to(A.device)                        # (L, 2, S)
    b2 = torch.stack(all_layer_biases_resid_2).to(A.device)
    L, B, S, _ = A.shape
    eye = torch.eye(S, device=A.device).expand(1, B, S, S)
    r1 = A * b1[:, 1].reshape(L, 1, 1, S) + eye * torch.diag_embed(b1[:, 0]).reshape(A.shape)
    if ablate == 0:
        ratio = torch.nn.functional.normalize(b2[:, 1].reshape(L, 1, 1, S) / b2[:, 0].reshape(L, 1, 1, S), p=1, dim=-1)
        r2 = torch.diag_embed(ratio.reshape(L, S)).reshape(A.shape) * torch.diag_embed(b2[:, 1]).reshape(A.shape) + \
            eye * torch.diag_embed(b2[:, 0]).reshape(A.shape)
        aug = r1 @ r2
    elif ablate == 1:
        aug = r1
    else:
        raise ValueError("ablate must be 0 or 1")
    aug = aug / aug.sum(dim=-1, keepdim=True)
    joint = aug[0]
    for i in range(1, L):
        joint = aug[i].bmm(joint)
    return joint, aug


def _head_means(model):
    return [(blk.attn.get_attention_map().sum(dim=1) / blk.attn.get_attention_map().shape[1]).detach() for blk in model.blocks]


def _grid(v):
    side = int(np.sqrt(v.shape[-1]))
    return v.reshape(-1, side, side)


def _generate_naive_rollout(self, input, start_layer=0, device=None):
    """(reference Baselines.generate_naive_rollout :181-194)"""
    with torch.no_grad():
        self.model(input if device is None else input.to(device))
    layers = _head_means(self.model)
    rollout, mats = compute_rollout_naive(layers, start_layer=start_layer)
    return _grid(rollout[:, 0, 1:]), mats, torch.stack(layers)


def _generate_rollout(self, input, InFlow=False, start_layer=0, device=None):
    """Attention rollout (reference Baselines.generate_rollout :196-240).  InFlow=True weighs every block's attention against
    the identity by the norms of its residual-stream branches (`compute_RAVE`); it reads the hooks of the reference's timm-based
    twin (`blk.get_input()`, `.get_input_plus_attn()`, `.get_mlp_val()`, `.attn.get_output()`, ViT_new_timm.py:223-312), which the
    build's ViT keeps too."""
    with torch.no_grad():
        self.model(input if device is None else input.to(device))
    layers = _head_means(self.model)
    if InFlow:
        shares = [_residual_shares(blk) for blk in self.model.blocks]
        rollout, mats = compute_RAVE(layers, [s[0] for s in shares], [s[1] for s in shares])
    else:
        rollout, mats = compute_rollout_attention(layers, start_layer=start_layer)
    return _grid(rollout[:, 0, 1:]), mats, torch.stack(layers)


def _generate_transition_attention_maps(self, input, target_class, start_layer=0, steps=20, with_integral=True,
                                        first_state=False, device="cuda:0"):
    """Transition Attention Maps (reference :306-357): Markov-chain states over the blocks times the
    integrated attention gradient of the last block -- the same K2-shaped accumulation as Baselines.IG.
    Returns (states, W_state, final, last head-mean CLS attention, last-step attention gradient)."""
    dev = hip_device(device)
    x = input.to(dev, torch.float32)
    x0 = x.detach().requires_grad_(True)
    out = self.model(x0, register_hook=True)
    _backward(out[0][target_class].sum())
    blocks = self.model.blocks
    maps = [blk.attn.get_attention_map().detach() for blk in blocks]
    grad0 = blocks[-1].attn.get_attn_gradients()
    b, h, s, _ = maps[-1].shape
    states = maps[-1].mean(1)[:, 0, :].reshape(b, 1, s)
    for i in range(start_layer, len(blocks))[::-1]:
        attn = maps[i].mean(1)
        states = torch.einsum('biw, bwh->h', states, attn).reshape(b, 1, s) + states
    alphas = torch.from_numpy(np.linspace(0, 1, steps)).to(dev, torch.float32)
    scaled = (x * alphas.reshape(steps, 1, 1, 1)).detach().requires_grad_(True)
    output = self.model(scaled, register_hook=True)
    _backward(output[:, target_class].sum())
    g = blocks[-1].attn.get_attn_gradients()                                   # (steps, h, s, s)
    last_map = blocks[-1].attn.get_attention_map()[-1:].detach()
    if with_integral:
        cls_rows = g[:, :, 0, :].contiguous().reshape(1, steps, h, s)
        ones = torch.ones((1, h, s), dtype=torch.float32, device=dev)
        W_state = K.ig_accum(cls_rows, ones, 0.0)[0].clamp(min=0).mean(0).reshape(b, 1, s)
    else:
        W_state = grad0.clamp(min=0).mean(1)[:, 0, :].reshape(b, 1, s)
    if first_state:
        states = maps[-1].mean(1)[:, 0, :].reshape(b, 1, s)
    final = states * W_state
    return (_grid(states[:, 0, 1:]), _grid(W_state[:, 0, 1:]), _grid(final[:, 0, 1:]), last_map.mean(1)[0, 0, 1:], g[-1:])


Baselines.generate_naive_rollout = _generate_naive_rollout
Baselines.generate_rollout = _generate_rollout
Baselines.generate_transition_attention_maps = _generate_transition_attention_maps


def _attn_attr(self, input, target_class, start_layer=0, device="cuda:0"):
    """Attention attribution rollout (reference Baselines.attn_attr :389-414)."""
    dev = hip_device(device)
    x0 = input.to(dev, torch.float32).detach().requires_grad_(True)
    out = self.model(x0, register_hook=True)
    _backward(out[0][target_class].sum())
    blocks = self.model.blocks
    b, h, s, _ = blocks[-1].attn.get_attention_map().shape
    states = blocks[-1].attn.get_attention_map().detach().mean(1)[:, 0, :].reshape(b, 1, s)
    for i in range(start_layer, len(blocks) - 1)[::-1]:
        states = states.bmm(blocks[i].attn.get_attention_map().detach().mean(1)) + states
    W_state = blocks[-1].attn.get_attn_gradients().clamp(min=0).mean(1)[:, 0, :].reshape(b, 1, s)
    return _grid((states * W_state)[:, 0, 1:])


def _bidirectional(self, input, target_class, steps=20, start_layer=4, samples=20, noise=0.2, mae=False, dino=False, ssl=False,
                   InFlow=False, device="cuda"):
    """Bidirectional transformer attribution (reference Baselines.bidirectional :417-518):
    head-importance-weighted attention rollout R (InFlow=True: chained through the residual-share matrices of :447-464 instead
    of R + cam R), times the integrated attention gradient of the last block.
    The integral needs every row of the (heads, S, S) gradient, so all `steps` gradients are reduced by
    xai_ig_accum_f32 as one [steps][heads*S*S] buffer."""
    dev = hip_device(device)
    x = input.to(dev, torch.float32)
    x0 = x.detach().requires_grad_(True)
    out = self.model(x0, register_hook=True)
    _backward(out[0][target_class].sum())
    blocks = self.model.blocks
    b, num_head, num_tokens, _ = blocks[-1].attn.get_attention_map().shape
    R = torch.eye(num_tokens, num_tokens, device=dev).expand(b, num_tokens, num_tokens)
    for nb, blk in enumerate(blocks):
        if nb < start_layer - 1:
            continue
        grad = blk.attn.get_attn_gradients()
        grad = grad.reshape(-1, grad.shape[-2], grad.shape[-1])
        cam = blk.attn.get_attention_map().detach()
        cam = cam.reshape(-1, cam.shape[-2], cam.shape[-1])
        Ih = torch.mean(torch.matmul(cam.transpose(-1, -2), grad).abs(), dim=(-1, -2))
        Ih = Ih / torch.sum(Ih)
        cam = torch.matmul(Ih, cam.reshape(num_head, -1)).reshape(num_tokens, num_tokens)
        if not InFlow:
            R = R + torch.matmul(cam, R)
            continue
        b1, b2 = _residual_shares(blk)                                                  # (2, S) each
        S = num_tokens
        r1 = cam * b1[1].reshape(1, 1, S) + R * torch.diag_embed(b1[0]).reshape(1, S, S)
        ratio = torch.nn.functional.normalize(b2[1].reshape(1, 1, S) / b2[0].reshape(1, 1, S), p=1, dim=-1)
        r2 = torch.diag_embed(ratio.reshape(S)).reshape(1, S, S) * torch.diag_embed(b2[1]).reshape(1, S, S) + \
            R * torch.diag_embed(b2[0]).reshape(1, S, S)
        R = r1 @ r2
    if ssl:
        if mae:
            return R[:, 1:, 1:].abs().mean(axis=1)
        if dino:
            return R[:, 1:, 1:].abs().mean(axis=1) + R[:, 0, 1:].abs()
        return R[:, 0, 1:].abs()
    alphas = torch.from_numpy(np.linspace(0, 1, steps)).to(dev, torch.float32)
    scaled = (x * alphas.reshape(steps, 1, 1, 1)).detach().requires_grad_(True)
    output = self.model(scaled, register_hook=True)
    _backward(output[:, target_class].sum())
    g = blocks[-1].attn.get_attn_gradients().contiguous()                      # (steps, heads, S, S)
    n = num_head * num_tokens * num_tokens
    ones = torch.ones((1, 1, n), dtype=torch.float32, device=dev)
    mean = K.ig_accum(g.reshape(1, steps, 1, n), ones, 0.0).reshape(num_head, num_tokens, num_tokens)
    W_state = mean.clamp(min=0).mean(0).reshape(b, num_tokens, num_tokens)
    attr = W_state * R
    if mae:
        return attr[:, 1:, 1:].mean(axis=1)
    if dino:
        return attr[:, 1:, 1:].mean(axis=1) + attr[:, 0, 1:]
    return _grid(attr[:, 0, 1:]), _grid(R[:, 0, 1:])


Baselines.attn_attr = _attn_attr
Baselines.bidirectional = _bidirectional
