"""Reference module path `util.model_utils` (reference util/model_utils.py:4-58): the three
single-image helpers the harness uses to pick classes.  Pure classifier calls -- PyTorch-ROCm."""
import torch


def _logits(model, input, device):
    out = model(input.to(device))
    return out if isinstance(out, torch.Tensor) else out.logits


def getPrediction(input, model, device, target_class):
    """-> (softmax probability, logit) of `target_class` (-1: the top class) as numpy scalars."""
    output = _logits(model, input, device)
    k = torch.max(output, 1)[1][0] if target_class == -1 else target_class
    prob = torch.nn.functional.softmax(output, dim=1)[0][k]
    return prob.detach().cpu().numpy(), output[0][k].detach().cpu().numpy()


def getClass(input, model, device, k=0):
    """index of the (k+1)-th highest logit."""
    output = _logits(model, input, device)
    if k == 0:
        return torch.max(output, dim=1)[1][0]
    return torch.topk(output, k + 1, dim=1)[1].squeeze()[k]


def getGradients(input, model, device, target_class):
    """d logit[target] / d input of one image -> (C,H,W)."""
    input = input.to(device)
    input.requires_grad = True
    score = model(input)[0][target_class]
    gradients = torch.autograd.grad(score, input)[0][0]
    input.requires_grad = False
    return gradients
