"""Autograd-aware collectives for the Hybrid-OP (tensor-parallel) layout, on torch.distributed (backend "nccl" = RCCL over
xGMI on MI355X; "gloo" in the CPU tests).  Same public names and forward/backward pairing as the reference's
src/UCF_VIT/utils/dist_functions.py (used ones: F_Identity_B_AllReduce:403, F_AllReduce_B_Identity:515,
F_Identity_B_Broadcast:501, F_Identity_B_AllReduce_VariableMapping:421), restated with one generic Function.

On MI355X the tensor-parallel messages are [B, N, D] activations: RCCL runs them on its own HIP stream; the waits below
only order the compute stream after the collective (no host synchronisation on the nccl backend).
"""
import torch
import torch.distributed as dist
from torch.autograd import Function

ReduceOp = dist.ReduceOp


def _all_reduce(t, op, group):
    t = t.contiguous().clone()
    dist.all_reduce(t, op=op, group=group)
    return t


def _broadcast(t, src, group):
    t = t.contiguous().clone()
    dist.broadcast(t, src, group=group)
    return t


class _Pair(Function):
    """forward: fwd_fn(tensor) ; backward: bwd_fn(grad).  fwd_fn / bwd_fn are closures over (op, src, group)."""

    @staticmethod
    def forward(ctx, fwd_fn, bwd_fn, tensor):
        ctx.bwd_fn = bwd_fn
        return fwd_fn(tensor)

    @staticmethod
    def backward(ctx, grad_output):
        return None, None, ctx.bwd_fn(grad_output)


def _ident(t):
    return t.view_as(t)


def F_Identity_B_AllReduce(tensor, group=dist.group.WORLD):
    """entry of a tensor-parallel region: identity forward, SUM all-reduce of the gradient backward"""
    return _Pair.apply(_ident, lambda g: _all_reduce(g, ReduceOp.SUM, group), tensor)


F_Identity_B_AllReduce_VariableMapping = F_Identity_B_AllReduce


def F_AllReduce_B_Identity(tensor, op=ReduceOp.SUM, group=dist.group.WORLD):
    """exit of a tensor-parallel region: all-reduce forward, identity backward"""
    return _Pair.apply(lambda t: _all_reduce(t, op, group), lambda g: g.clone(), tensor)


def F_Identity_B_Broadcast(tensor, src, group=dist.group.WORLD):
    """identity forward; backward broadcasts the gradient of rank `src` to the group"""
    return _Pair.apply(_ident, lambda g: _broadcast(g, src, group), tensor)


def F_Broadcast_B_Reduce(tensor, src, group=dist.group.WORLD):
    """broadcast forward; backward reduces (SUM) the gradients onto `src`, zero elsewhere"""
    def bwd(g):
        g = g.contiguous().clone()
        dist.reduce(g, src, op=ReduceOp.SUM, group=group)
        if dist.get_rank() != src:
            g.zero_()
        return g
    return _Pair.apply(lambda t: _broadcast(t, src, group), bwd, tensor)


def F_AllReduce_B_AllReduce(tensor, op=ReduceOp.SUM, group=dist.group.WORLD):
    return _Pair.apply(lambda t: _all_reduce(t, op, group), lambda g: _all_reduce(g, op, group), tensor)


def all_gather_last_dim(tensor, group=dist.group.WORLD):
    """all-gather along the last dim forward; backward keeps this rank's slice of the gradient"""
    world, rank = dist.get_world_size(group), dist.get_rank(group)

    def fwd(t):
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous(), group=group)
        return torch.cat(parts, dim=-1)

    def bwd(g):
        w = g.shape[-1] // world
        return g[..., rank * w:(rank + 1) * w].contiguous()
    return _Pair.apply(fwd, bwd, tensor)
