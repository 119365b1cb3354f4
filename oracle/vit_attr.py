"""Oracle: attention-space IG of a hooked ViT (test infrastructure only).
Restates Baselines.IG, util/attribution_methods/VIT_LRP/ViT_explanation_generator.py:358-386:
sequential passes on input*alpha, full (heads,S,S) gradient sum, /steps, clamp, head mean, CLS row."""
import numpy as np
import torch


def attention_ig(model, x, target, steps=20):
    dev = next(model.parameters()).device
    total = None
    for alpha in np.linspace(0, 1, steps):
        scaled = (torch.from_numpy(np.asarray(x, dtype=np.float32)).to(dev) * alpha).requires_grad_(True)
        out = model(scaled, register_hook=True)
        out[0][int(target)].sum().backward()
        g = model.blocks[-1].attn.get_attn_gradients().detach().cpu().numpy()
        total = g.copy() if total is None else total + g
    w = np.maximum(total / np.float32(steps), 0).mean(axis=1)[:, 0, :]
    side = int(np.sqrt(w.shape[-1] - 1))
    return w[:, 1:].reshape(-1, side, side)


# ---- InFlow rollout (compute_RAVE :48-88, generate_rollout(InFlow=True) :196-240), NumPy float32 on activations read from the hooks
def _shares(a, b):
    """2-norm per token of the two branches of a residual addition, L1-normalised over the pair [:214-235]."""
    na = np.sqrt((a.astype(np.float32) ** 2).sum(axis=1, dtype=np.float32))
    nb = np.sqrt((b.astype(np.float32) ** 2).sum(axis=1, dtype=np.float32))
    tot = np.maximum(np.abs(na) + np.abs(nb), np.float32(1e-12))
    return na / tot, nb / tot


def inflow_rollout(model, x):
    """-> (rollout (1, side, side), row-normalised per-block matrices (L, 1, S, S)); one image."""
    dev = next(model.parameters()).device
    with torch.no_grad():
        model(torch.from_numpy(np.asarray(x, dtype=np.float32)).to(dev))
    mats = []
    for blk in model.blocks:
        A = blk.attn.get_attention_map().detach().cpu().numpy()[0]
        A = A.sum(axis=0, dtype=np.float32) / np.float32(A.shape[0])                     # head mean as sum / heads [:210]
        S = A.shape[-1]
        inp, att = blk.get_input().detach().cpu().numpy()[0], blk.attn.get_output().detach().cpu().numpy()[0]
        res, mlp = blk.get_input_plus_attn().detach().cpu().numpy()[0], blk.get_mlp_val().detach().cpu().numpy()[0]
        in_share, attn_share = _shares(inp, att)
        res_share, mlp_share = _shares(res, mlp)
        r1 = A * attn_share[None, :] + np.eye(S, dtype=np.float32) * np.diag(in_share)      # [:67]
        ratio = mlp_share / res_share
        ratio = ratio / np.maximum(np.abs(ratio).sum(dtype=np.float32), np.float32(1e-12))   # F.normalize(p=1) [:70-71]
        r2 = np.diag(ratio) * np.diag(mlp_share) + np.eye(S, dtype=np.float32) * np.diag(res_share)   # [:73]
        m = r1 @ r2
        mats.append(m / m.sum(axis=-1, keepdims=True))                                        # [:82]
    joint = mats[0]
    for m in mats[1:]:
        joint = m @ joint                                                                     # [:84-87]
    side = int(np.sqrt(joint.shape[-1] - 1))
    return joint[0, 1:].reshape(1, side, side), np.stack(mats)[:, None]
