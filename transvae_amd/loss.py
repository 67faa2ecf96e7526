"""Loss functions with the reference's names (transvae/loss.py), computed by the fused HIP kernel."""
import torch

from . import ops


def trans_vae_loss(x, x_out, mu, logvar, true_len, pred_len, true_prop, pred_prop, weights, beta=1, pad_idx=None):
    """Weighted token cross-entropy + length cross-entropy + beta * KL (+ property MSE).

    Same signature and return order as the reference (transvae/loss.py:23-38):
    ``(total, BCEmol, BCEmask, KLD, MSE)``.  ``x`` are the (B, 127) source token ids (float or
    long); ``true_len`` (B,) or (B,1) are the length classes (the reference passes
    src_mask.sum(-1), trans_models.py:219); None lets the kernel count the non-pad tokens itself
    (``pad_idx``, default len(weights) - 1 = the '_' id of the shipped vocabularies).
    All five values stay on the device; gradients for x_out, mu, logvar, pred_len and pred_prop
    are produced in the same pass and handed to autograd.
    """
    tokens = x.long()
    if pad_idx is None:
        pad_idx = weights.shape[0] - 1
    return ops.LossFn.apply(tokens, x_out, mu, logvar, pred_len, true_prop, pred_prop,
                            weights.to(torch.float32), float(beta), pad_idx, true_len)


def vae_loss(x, x_out, mu, logvar, true_prop, pred_prop, weights, beta=1):
    """Weighted token cross-entropy + beta * KL (+ property MSE): the loss of the RNN models, same signature and return
    order as the reference (transvae/loss.py:8-21): ``(total, BCE, KLD, MSE)``.  The same fused kernel as trans_vae_loss,
    launched without a length head."""
    total, bce, _, kld, mse = ops.LossFn.apply(x.long(), x_out, mu, logvar, None, true_prop, pred_prop,
                                               weights.to(torch.float32), float(beta), weights.shape[0] - 1, None)
    return total, bce, kld, mse
