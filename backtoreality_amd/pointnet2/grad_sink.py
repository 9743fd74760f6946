"""Flat gradient sinks for steps that run the same parameters through several forwards.

A Back-to-Reality step (train_Votenet_BR.py:267-289, train_GF_BR.py:330-356) sends a source and a
target batch through ONE model and calls backward once.  Every native node of this package
(whole backbone, point-wise chain, fused set-abstraction layer) already writes the gradients of
its parameters into one flat buffer and hands autograd ~20 - 80 views of it.  With two forwards,
autograd's input buffers then add the two branches' contributions parameter by parameter: 117
element-wise launches per VoteNet BR step (profiles/r06_h_br_kernel_stats.md: 0.35 ms of kernel
time plus their launch gaps on a 8.4 ms step), each over a few hundred floats.

Inside `with grad_sink.scope():` a native node takes one more input, a flat leaf tensor of the
size of its gradient buffer (the SINK, one per node and shape, values never read), and returns
the whole buffer as that input's gradient and None for the parameters.  Autograd adds the two
branches' buffers with ONE launch per node, accumulates into the sink, and the sink's
post-accumulate hook hands every parameter its view as `.grad` (added to an existing `.grad`).
The sums are the same element-wise sums: `p.grad` is bit-identical (tests/test_grad_sink_gpu.py).

Not on by default: parameter hooks never see these gradients (torch's DistributedDataParallel
relies on them: train.py keeps the sinks off under that wrapper; FlatGradParallel reads `.grad`
after the backward and is fine), and `torch.autograd.grad(..., parameters)` returns None for
them.  The training steps that run two forwards switch it on."""
import torch

_DEPTH = [0]


class scope(object):
    def __enter__(self):
        _DEPTH[0] += 1
        return self

    def __exit__(self, *exc):
        _DEPTH[0] -= 1
        return False


def active():
    """True inside a scope, with gradients on, outside a HIP-graph capture (a replayed step keeps
    static `.grad` tensors; the hook hands out fresh views)."""
    return _DEPTH[0] > 0 and torch.is_grad_enabled() and \
        not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())


def all_leaves(params):
    """A sink stands for PARAMETERS: a node one of whose weight operands was computed (a
    concatenation of several heads' weights, a tied / transformed weight) keeps returning
    per-operand gradients -- autograd has to carry them further."""
    return all(p is None or (p.is_leaf and p.requires_grad) for p in params)


class Sink(object):
    """`views_of(flat)`: the per-parameter gradients (views of `flat`, None where a parameter
    gets none), aligned with `params`."""

    def __init__(self, nfloats, device, views_of):
        self.tensor = torch.zeros((max(int(nfloats), 1),), dtype=torch.float32, device=device,
                                  requires_grad=True)
        self.params = ()
        self.views_of = views_of
        self.tensor.register_post_accumulate_grad_hook(self._distribute)

    def bind(self, params):
        """The parameters of the call being made (they may have been replaced since the last)."""
        self.params = tuple(params)
        return self.tensor

    def _distribute(self, t):
        flat, t.grad = t.grad, None
        if flat is None:
            return
        with torch.no_grad():
            for p, v in zip(self.params, self.views_of(flat)):
                if p is None or v is None or not (p.requires_grad and p.is_leaf):
                    continue   # (a computed operand: its node returns that gradient itself)
                if not v.is_contiguous():   # (a first layer behind a padded input width)
                    v = v.contiguous()
                p.grad = v if p.grad is None else p.grad + v
