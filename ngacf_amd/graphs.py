"""hipGraph capture of the launch-bound loops (MI355X-first: HIP graphs instead of a tracing compiler).

A reference-semantics BPR step is ~180 small launches (2 propagations x 3 stages x {transform, traversal,
hub-row pass} forward and backward, scoring, loss, per-head gradient plumbing, Adam); eager mode is bound by the
host's launch rate, not by the GPU.  `GraphedBPRStep` captures one whole step -- forward x2, loss, backward,
optimizer -- into one graph and replays it; dropout seeds come from a device-side counter that a captured add
advances, so every replay draws fresh masks.  `GraphedPropagate` does the same for the eval-mode propagation.
"""
import torch

from .SPUIGACF import _GATBase

# the warm-up runs on a side stream by design; autograd's stream-mismatch warning is expected there
try:
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
except AttributeError:      # older torch
    pass


def _gats(model):
    return [m for m in model.modules() if isinstance(m, _GATBase)]


class GraphedBPRStep:
    """step(users, pos, neg) -> loss (a device scalar, valid until the next call).

    optim must be capturable (torch.optim.Adam(..., capturable=True)); batch size is fixed at capture."""

    def __init__(self, model, lossfn, optim, adj, batch_size, warmup=3, shared_propagation=False):
        dev = next(model.parameters()).device
        self.model, self.optim = model, optim
        for g in _gats(model):
            g.use_device_seed(dev, True)
        self.upn = torch.zeros(3, batch_size, dtype=torch.long, device=dev)     # users / positives / negatives
        self.u, self.p, self.n = self.upn[0], self.upn[1], self.upn[2]
        model.train()
        gats = _gats(model)

        def step():
            for g in gats:
                g.advance_device_seed()                     # one captured add per step: fresh masks every replay
            optim.zero_grad(set_to_none=True)
            if shared_propagation:                          # opt-in: one propagation per step (model.forward_pairs)
                loss = lossfn(*model.forward_pairs(self.u, self.p, self.n, adj))
            else:                                           # reference semantics: two forwards, two propagations
                loss = lossfn(model(self.u, self.p, adj), model(self.u, self.n, adj))
            loss.backward()
            optim.step()
            return loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the capture: plans, caches, allocator
            for _ in range(warmup):
                step()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optim.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = step()

    def __call__(self, users, pos=None, neg=None):
        """step(users, pos, neg), or step(upn) with the three index rows packed as one [3, B] tensor (one copy)."""
        if pos is None:
            self.upn.copy_(users)
        else:
            self.u.copy_(users)
            self.p.copy_(pos)
            self.n.copy_(neg)
        self.graph.replay()
        return self.loss


class GraphedPropagate:
    """propagate() -> the [N, d] propagated embeddings of the model in eval mode (static output buffer)."""

    def __init__(self, model, adj, warmup=2):
        self.model = model
        model.eval()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                model.propagate(adj)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = model.propagate(adj)

    def __call__(self):
        self.graph.replay()
        return self.out
