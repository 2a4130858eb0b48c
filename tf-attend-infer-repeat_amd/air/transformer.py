"""Spatial transformer op -- mirror of /root/reference/air/transformer.py:18-175.

``transformer(U, theta, out_size)`` keeps the reference signature; U is
[B, H, W, 1] (or [B, H, W]) float32 on the GPU, theta [B, 6] or [B, 2, 3].
Runs the hand-written HIP kernels (air_transformer_fwd / air_transformer_bwd); no CPU fallback.
When U or theta require a gradient the op is differentiable the way the reference's is under
tf.gradients (same op order, see include/air_hip.h); torch.autograd only carries the call."""
import ctypes as C

import torch

from . import _hip as H


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def transformer_grad(U, theta, out_size, d_out, need_dU=True, need_dtheta=True):
    """(d_U [B,Hi,Wi], d_theta [B,6]) of transformer(U, theta, out_size) for an incoming d_out [B,Ho,Wo]."""
    B, Hi, Wi = int(U.shape[0]), int(U.shape[1]), int(U.shape[2])
    Ho, Wo = int(out_size[0]), int(out_size[1])
    Uc = U.reshape(B, Hi, Wi).contiguous().float()
    th = theta.reshape(B, 6).contiguous().float()
    g = d_out.reshape(B, Ho, Wo).contiguous().float()
    dU = torch.empty_like(Uc) if need_dU else None
    dth = torch.empty_like(th) if need_dtheta else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
    H.check(H.lib().air_transformer_bwd(p(Uc), p(th), p(g), p(dU), p(dth), B, Hi, Wi, Ho, Wo, _stream(U.device)),
            "air_transformer_bwd")
    return dU, dth


class _TransformerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, theta, out_size):
        ctx.save_for_backward(U, theta)
        ctx.out_size = out_size
        return transformer(U.detach(), theta.detach(), out_size)

    @staticmethod
    def backward(ctx, d_out):
        U, theta = ctx.saved_tensors
        dU, dth = transformer_grad(U, theta, ctx.out_size, d_out, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return (dU.reshape(U.shape) if dU is not None else None,
                dth.reshape(theta.shape) if dth is not None else None, None)


def transformer(U, theta, out_size, name="SpatialTransformer", **kwargs):
    if torch.is_grad_enabled() and (U.requires_grad or theta.requires_grad):
        return _TransformerFn.apply(U, theta, tuple(out_size))
    if not U.is_cuda:
        raise H.AirHipError("transformer: U must be a device tensor (no CPU fallback)")
    squeeze = U.dim() == 4
    if squeeze and U.shape[3] != 1:
        raise NotImplementedError("only num_channels == 1 is on the AIR path (air_model.py:331, 364)")
    B, Hi, Wi = int(U.shape[0]), int(U.shape[1]), int(U.shape[2])
    Ho, Wo = int(out_size[0]), int(out_size[1])
    Uc = U.reshape(B, Hi, Wi).contiguous().float()
    th = theta.reshape(B, 6).contiguous().float()
    out = torch.empty(B, Ho, Wo, dtype=torch.float32, device=U.device)
    s = C.c_void_p(torch.cuda.current_stream(U.device).cuda_stream)
    H.check(H.lib().air_transformer_fwd(C.c_void_p(Uc.data_ptr()), C.c_void_p(th.data_ptr()),
                                        C.c_void_p(out.data_ptr()), B, Hi, Wi, Ho, Wo, s), "air_transformer_fwd")
    return out.unsqueeze(3) if squeeze else out
