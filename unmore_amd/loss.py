"""The stage-1 training loss (reference: train_objectness_net.py:215-254) as one fused
HIP kernel pair: value and gradient w.r.t. both prediction maps in a single pass."""
import torch

from . import ops


class _LossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pc, ps, gc, gs, sal, cfg):
        need = pc.requires_grad or ps.requires_grad
        out5, dpc, dps = ops.objectness_loss(pc.contiguous(), ps.contiguous(), gc.contiguous(), gs.contiguous(),
                                             sal.contiguous() if sal is not None else None, *cfg, need_grad=True)
        ctx.save_for_backward(dpc, dps)
        ctx.terms = out5
        return out5[0].clone(), out5[1:].clone()

    @staticmethod
    def backward(ctx, g_total, g_terms):
        dpc, dps = ctx.saved_tensors
        # d(total)/d(pred) was produced in the forward pass; the incoming scalar (1 for loss.backward()) is applied on the
        # device -- reading it on the host would synchronise the stream every step
        g = g_total.detach().reshape(1).to(torch.float32).contiguous()
        return ops.scale_by_device_scalar(dpc, g), ops.scale_by_device_scalar(dps, g), None, None, None, None


def objectness_loss(out_dict, gt_center_fields, gt_sdf_maps, gt_saliency_maps, center_field_loss_type="l2",
                    sdf_loss_type="l1", use_sdf_gradient_loss=True, use_sdf_binary_mask_loss=True, return_terms=False):
    """loss = mean((pc-gc)^2 | abs) + mean(abs(ps-gs) | ^2) [+ mean over the (H-1)x(W-1) forward-difference
    maps] [+ BCE(sigmoid(ps), saliency)] with unit weights, as the reference's flags select."""
    for t in (center_field_loss_type, sdf_loss_type):
        if t not in ("l1", "l2"):
            raise NotImplementedError
    cfg = (center_field_loss_type == "l2", sdf_loss_type == "l2", bool(use_sdf_gradient_loss), bool(use_sdf_binary_mask_loss))
    total, terms = _LossFunction.apply(out_dict["center_fields"].float(), out_dict["sdf_maps"].float(), gt_center_fields.float(),
                                       gt_sdf_maps.float(), gt_saliency_maps.float() if gt_saliency_maps is not None else None, cfg)
    return (total, terms) if return_terms else total
