"""Autograd glue of the HIP engine: ``GssdTrainFn`` = HIP forward plan + HIP backward plan (gssd/backward.py).

There is ONE backend.  The backward runs against the plan instance that produced the forward: the plan and its run
counter are saved in ``ctx``; a plan whose forward still awaits its backward is marked busy, so a second forward (another
micro-batch, an evaluation pass, a bench survey) takes another plan instance with its own activation buffers instead of
overwriting them, and a backward that finds its plan's counter changed raises instead of using stale activations.

Parameter gradients.  The backward plan writes every gradient into a 16-byte aligned slice of one flat fp32 tensor.  By
default those slices are RETURNED to autograd like any Function's gradients (``torch.autograd.grad``, tensor hooks, non-leaf
parameters such as ``nn.DataParallel`` replicas and DDP's hooks all see them).  When the call is a plain ``.backward()`` on
leaf parameters without tensor hooks the slices are instead handed out as ``param.grad`` directly -- no AccumulateGrad clone
per parameter, and ``gssd.dist.allreduce_grads`` can reduce the flat tensor in place; ``net.direct_grad_handout = False``
switches that fast path off.
"""
import torch

from . import _lib


class _Lease:
    """Lifetime of one grad-enabled forward: while it is alive (= the autograd node is alive) the plan stays busy."""
    __slots__ = ('plan', 'gen')

    def __init__(self, plan):
        self.plan, self.gen = plan, plan.generation
        plan.busy = True

    def release(self):
        p = self.plan
        if p is not None and p.generation == self.gen:
            p.busy = False
        self.plan = None

    def __del__(self):
        self.release()


def _will_accumulate_all(ctx, params):
    """True when this backward pass is a plain ``.backward()``: every leaf parameter's AccumulateGrad node will run.
    (Inside ``torch.autograd.grad`` the engine refuses the query for leaf nodes -- RuntimeError -- which answers it too.)"""
    try:
        nodes = ctx.next_functions[-len(params):]          # tensor inputs only: (x, *params)
        will = torch._C._will_engine_execute_node
        if len(nodes) != len(params):
            return False
        for (node, _), p in zip(nodes, params):
            if not p.requires_grad:
                continue
            if node is None or type(node).__name__ != 'AccumulateGrad' or not will(node):
                return False
        return True
    except (AttributeError, RuntimeError):
        return False


class GssdTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        if x.requires_grad:
            raise _lib.GssdError('the HIP path provides no gradient with respect to the input images '
                                 '(x.requires_grad=True); detach the input')
        loc, conf, plan = net._engine.forward_plan(x, True, net.__dict__.get('_events'), bool(net.__dict__.get('_want_maps')))
        ctx.net, ctx.params = net, params
        ctx.plan, ctx.gen = plan, plan.generation
        ctx.lease = _Lease(plan)
        ctx.set_materialize_grads(False)
        return loc, conf

    @staticmethod
    def backward(ctx, dloc, dconf):
        net, params, plan = ctx.net, ctx.params, ctx.plan
        if plan.generation != ctx.gen:
            raise _lib.GssdError('GSSD HIP backward: the forward plan ran again after the forward this backward belongs to '
                                 '(its activation buffers were overwritten) -- backward twice through one forward, or a plan '
                                 'invalidated in between')
        if dloc is None:
            dloc = torch.zeros(plan.B, plan.P, 4, device=plan.dev)
        if dconf is None:
            dconf = torch.zeros(plan.B, plan.P, plan.nc, device=plan.dev)
        bwd = plan.backward_plan()
        direct = (net.__dict__.get('direct_grad_handout', True)
                  and all(p.is_leaf and not p._backward_hooks for p in params if p.requires_grad)
                  and _will_accumulate_all(ctx, params))
        if direct:
            # an existing gradient that still aliases the plan's flat buffer (from an earlier backward of this plan) is moved out
            # of the way first; then p.grad = slice (or += into a foreign gradient)
            lo = bwd.flat.data_ptr()
            hi = lo + bwd.flat.numel() * 4
            for p in params:
                if p.grad is not None and lo <= p.grad.data_ptr() < hi:
                    p.grad = p.grad.clone()
        # gssd.dist.OverlappedGradReducer all-reduces ranges of the flat buffer IN PLACE while the backward is still running.
        # That is only the gradient when every p.grad becomes its slice of that buffer: with accumulation into an existing
        # p.grad, tensor hooks, torch.autograd.grad or direct_grad_handout=False the slices are copied / added elsewhere
        # (possibly while RCCL is rewriting them), so the hook is withheld and the reducer told to reduce p.grad afterwards.
        hook = getattr(net._engine, 'grad_segment_hook', None)
        if hook is not None and not (direct and all(p.grad is None for p in params if p.requires_grad)):
            net._engine.grad_segment_skipped = True
            hook = None
        bwd.segment_hook = hook
        grads = bwd.run(dloc.contiguous(), dconf.contiguous())
        ctx.lease.release()
        if direct:
            for p, g in zip(params, grads):
                if g is None or not p.requires_grad:
                    continue
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)
            return (None, None) + (None,) * len(params)
        return (None, None) + tuple(g if (g is not None and p.requires_grad) else None for p, g in zip(params, grads))


class PixelLinkTrainFn(torch.autograd.Function):
    """PixelLink++ (pixel_link/model.py): HIP forward plan + gssd/backward.py::PixelLinkBackwardPlan.  Same contract as GssdTrainFn;
    the parameter gradients are returned to autograd (no direct hand-out: the PixelLink++ row has no data-parallel reducer)."""

    @staticmethod
    def forward(ctx, net, x, *params):
        if x.requires_grad:
            raise _lib.GssdError('the HIP path provides no gradient with respect to the input images '
                                 '(x.requires_grad=True); detach the input')
        out_1, out_2, plan = net._engine.forward_plan(x, True, net.__dict__.get('_events'))
        ctx.net, ctx.params = net, params
        ctx.plan, ctx.gen = plan, plan.generation
        ctx.lease = _Lease(plan)
        ctx.set_materialize_grads(False)
        return out_1, out_2

    @staticmethod
    def backward(ctx, d1, d2):
        params, plan = ctx.params, ctx.plan
        if plan.generation != ctx.gen:
            raise _lib.GssdError('PixelLink++ HIP backward: the forward plan ran again after the forward this backward belongs to')
        Ho = plan.H_out
        if d1 is None:
            d1 = torch.zeros(plan.B, 2, Ho, Ho, device=plan.dev)
        if d2 is None:
            d2 = torch.zeros(plan.B, 16, Ho, Ho, device=plan.dev)
        grads = plan.backward_plan().run(d1.contiguous().float(), d2.contiguous().float())
        ctx.lease.release()
        # the slices alias the plan's flat buffer, which the next backward of this plan overwrites: hand out copies
        return (None, None) + tuple(g.clone() if (g is not None and p.requires_grad) else None for p, g in zip(params, grads))
