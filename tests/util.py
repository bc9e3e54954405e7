import numpy as np
import torch


def rel_err(a, b):
    """max-norm relative error |a-b|_inf / |b|_inf (b = reference)."""
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).double().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).double().cpu()
    den = b.abs().max().item()
    return (a - b).abs().max().item() / (den if den > 0 else 1.0)


def l2_rel(a, b):
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).double().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).double().cpu()
    den = b.norm().item()
    return (a - b).norm().item() / (den if den > 0 else 1.0)
