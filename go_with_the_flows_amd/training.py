"""One training step of the flow-mixture model as a single hipGraph replay.

The reference's loop (lib/networks/training.py:25-60: forward, loss, backward, optimiser step per batch) is host-bound on
an MI355X: ~3000 kernel launches per step for the airplane config.  ``GraphedTrainStep`` captures forward + loss + backward
once and replays it per batch (airplane config, K=4 x 33 couplings, 64 x 2048 points: see bench.py's also.train_step); the
optimiser stays outside the graph because its bias corrections and the learning-rate schedule are host-side state.

Data-parallel runs (one process per GPU, the model converted to SyncBatchNorm as train_ae.py:152 does): the SAME single graph
per rank.  Every exchange is inside it -- the packed BatchNorm-statistic all-reduces of the decoders' phase-split pipeline
(4 per depth level), the encoder's, the row all-gathers of the per-shape modules, and the gradient exchange
(dist.OverlappedGradients: one asynchronous all-reduce per decoder launched from inside the backward pass + one flat remainder)
-- as RCCL kernels captured with the rest (RCCL collectives are stream-capturable); no host synchronisation anywhere in the
step.  This replaces DistributedDataParallel's bucketed all-reduce (train_ae.py:153), whose hooks are host-driven.
"""
import torch

from .dist import graph_capture as _graph_capture


class GraphedTrainStep:
    """step = GraphedTrainStep(model, criterion, optimizer, g_example, p_example); loss, pnll, gnll, gent = step(g, p).

    * inputs are copied into static buffers, so every batch must have the example's shape (drop the ragged last batch,
      as the reference's DataLoader does with drop_last=True, train_ae.py:97);
    * the returned loss terms are static tensors overwritten by the next call -- ``float()`` / ``.item()`` them first;
    * BatchNorm running statistics, the reparameterisation noise (graph-safe Philox) and ``.grad`` live inside the graph;
      parameters are updated in place by ``optimizer.step()`` after each replay.
    """

    def __init__(self, model, criterion, optimizer, g_example, p_example, warmup=False, warmup_iters=2, data_parallel=None,
                 average_gradients=True):
        from .dist import OverlappedGradients, sharded
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        # data_parallel=None: decided by the process group (more than one rank, or GWTF_FORCE_SHARDED=1)
        self.reducer = OverlappedGradients(model, average=average_gradients) if (sharded() if data_parallel is None else data_parallel) else None
        self.g_static, self.p_static = g_example.clone(), p_example.clone()
        self.use_warmup_weights = warmup
        side = torch.cuda.Stream(device=g_example.device)
        side.wait_stream(torch.cuda.current_stream(g_example.device))
        # The warm-up iterations run forward + backward with no optimiser step in between: they must not leave a trace in the
        # model (BatchNorm running statistics / num_batches_tracked) or in the random-number stream, or the state after
        # construction would differ from the reference loop's (training.py:25-60) by `warmup_iters` phantom batches.
        bn_state = [(b, b.detach().clone()) for b in model.buffers()]
        rng_cpu, rng_dev = torch.get_rng_state(), torch.cuda.get_rng_state(g_example.device)
        with torch.cuda.stream(side):
            for _ in range(warmup_iters):              # allocator + lazy initialisation outside the capture
                self._fwd_bwd()
            with torch.no_grad():
                for b, saved in bn_state:
                    b.copy_(saved)
        torch.cuda.current_stream(g_example.device).wait_stream(side)
        torch.set_rng_state(rng_cpu)
        torch.cuda.set_rng_state(rng_dev, g_example.device)
        for mod in model.modules():                    # restored buffers: drop every packed-weight cache keyed on them
            if hasattr(mod, 'invalidate_packed_weights'):
                mod.invalidate_packed_weights()
        # no autograd graph of an earlier iteration may be alive during capture (hipStreamEndCapture crashes otherwise)
        self.terms = None
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with _graph_capture(self.graph):
            self.terms = self._fwd_bwd()

    def _fwd_bwd(self):
        self.optimizer.zero_grad(set_to_none=True)
        enc, dec = self.model.forward_fused(self.g_static, self.p_static, self.use_warmup_weights)
        loss, pnll, gnll, gent = self.criterion.fused(enc, dec)
        if self.reducer is not None:
            with self.reducer:                         # gradients summed / averaged over the ranks, overlapped with the backward pass
                loss.backward()
        else:
            loss.backward()
        return tuple(t.detach() for t in (loss, pnll, gnll, gent))

    def __call__(self, g_input, p_input):
        self.g_static.copy_(g_input)
        self.p_static.copy_(p_input)
        self.graph.replay()
        self.optimizer.step()
        return self.terms
