"""One-process-per-GPU driver of the feature-sharded minibatch step (plumbing only: torch.distributed + streams).

    partial sums on the local shard  ->  ONE all-reduce (RCCL, or gloo staged through the host for tests)
                                     ->  multipliers / bias recurrence / scatter into the local shard

All three pieces are enqueued on ONE torch stream (a real, non-default stream: the library treats a NULL stream as
"use my own", which would not be ordered with torch's collectives), so no host synchronisation is needed inside a step.

pipeline=True overlaps the exchange with compute, the schedule every step being

    compute stream:   gather(b+1)                 | wait AR(b) | update(b)   | gather(b+2) ...
    RCCL stream   :   AR(b) .....................   AR(b+1) ................   AR(b+2) ...

i.e. the all-reduce of batch b+1 is issued as soon as its partial sums exist and runs under the update of batch b
(and under the gather of batch b+2); the RCCL stream is never idle, which is what matters when the step is bound by the
exchange (264 B per example per GPU).  The price: the sums of batch b+1 are gathered BEFORE the
update of batch b is applied -- "one batch stale", restated exactly by oracle fmo_sgd_epoch_minibatch_pipelined, so
this mode has a deterministic definition and its own parity tests.
"""
import torch
import torch.distributed as dist

from . import capi


class ShardedSGD:
    def __init__(self, handle, slot, n_rows, batch, w0_chunk=256, apply=capi.APPLY_DEFAULT, flags=capi.FLAG_BIAS_LAG,
                 backend="nccl", pipeline=False):
        self.h, self.slot, self.n_rows, self.batch = handle, slot, int(n_rows), int(batch)
        self.w0_chunk, self.apply, self.flags, self.backend = w0_chunk, apply, flags, backend
        self.pipeline = bool(pipeline)
        self.kp1 = handle.info().k_padded + 1
        self.stream = torch.cuda.Stream()
        # two exchange buffers: with FMX_FLAG_BIAS_LAG the recurrence of batch b reads its rest values while batch b+1
        # is already being gathered
        self.bufs = [torch.empty(self.batch * self.kp1, dtype=torch.float32, device="cuda") for _ in range(2)]
        self.step_no = 0

    def _exchange(self, view):
        """start the all-reduce of one partial buffer; returns a handle for _wait (None = already complete)"""
        if self.backend == "nccl":
            return dist.all_reduce(view, async_op=True) if self.pipeline else dist.all_reduce(view)
        host = view.cpu()                                   # gloo: stage through the host (testing the path without RCCL)
        dist.all_reduce(host)
        view.copy_(host)
        return None

    def epoch(self):
        """one pass over the slot's rows (fm_learn_sgd_element.h:56-67, restated batch rule)"""
        h, s = self.h, self.stream.cuda_stream
        starts = list(range(0, self.n_rows, self.batch))
        with torch.cuda.stream(self.stream):
            if not self.pipeline:
                for row0 in starts:
                    nb = min(self.batch, self.n_rows - row0)
                    view = self.bufs[self.step_no & 1][: nb * self.kp1]
                    h.sgd_partial(self.slot, row0, nb, view.data_ptr(), s)
                    self._exchange(view)
                    h.sgd_finish(self.slot, row0, nb, view.data_ptr(), self.apply, self.w0_chunk, s, self.batch, self.flags)
                    self.step_no += 1
                return
            views, works = {}, {}

            def gather(i):
                nb = min(self.batch, self.n_rows - starts[i])
                views[i] = self.bufs[i & 1][: nb * self.kp1]
                h.sgd_partial(self.slot, starts[i], nb, views[i].data_ptr(), s)
                works[i] = self._exchange(views[i])          # RCCL stream: waits for this gather, queues behind AR(i-1)
            gather(0)
            for i, row0 in enumerate(starts):
                if i + 1 < len(starts):
                    gather(i + 1)                            # reads the parameters before update(i): one batch stale
                if works[i] is not None:
                    works[i].wait()                          # compute stream waits for AR(i); no host block
                nb = min(self.batch, self.n_rows - row0)
                h.sgd_finish(self.slot, row0, nb, views[i].data_ptr(), self.apply, self.w0_chunk, s, self.batch, self.flags)
                del views[i], works[i]
                self.step_no += 1

    def synchronize(self):
        self.stream.synchronize()
        self.h.synchronize()
