"""One-process-per-GPU driver of the feature-sharded minibatch step (plumbing only: torch.distributed + streams).

    partial sums on the local shard  ->  ONE all-reduce (RCCL, or gloo staged through the host for tests)
                                     ->  multipliers / bias recurrence / scatter into the local shard

All three pieces are enqueued on ONE torch stream (a real, non-default stream: the library treats a NULL stream as
"use my own", which would not be ordered with torch's collectives), so no host synchronisation is needed inside a step.
"""
import torch
import torch.distributed as dist

from . import capi


class ShardedSGD:
    def __init__(self, handle, slot, n_rows, batch, w0_chunk=256, apply=capi.APPLY_DEFAULT, flags=capi.FLAG_BIAS_LAG,
                 backend="nccl"):
        self.h, self.slot, self.n_rows, self.batch = handle, slot, int(n_rows), int(batch)
        self.w0_chunk, self.apply, self.flags, self.backend = w0_chunk, apply, flags, backend
        self.kp1 = handle.info().k_padded + 1
        self.stream = torch.cuda.Stream()
        # two exchange buffers: with FMX_FLAG_BIAS_LAG the recurrence of batch b reads its rest values while batch b+1
        # is already being gathered
        self.bufs = [torch.empty(self.batch * self.kp1, dtype=torch.float32, device="cuda") for _ in range(2)]
        self.step_no = 0

    def epoch(self):
        """one pass over the slot's rows (fm_learn_sgd_element.h:56-67, restated batch rule)"""
        h, s = self.h, self.stream.cuda_stream
        with torch.cuda.stream(self.stream):
            for row0 in range(0, self.n_rows, self.batch):
                nb = min(self.batch, self.n_rows - row0)
                view = self.bufs[self.step_no & 1][: nb * self.kp1]
                h.sgd_partial(self.slot, row0, nb, view.data_ptr(), s)
                if self.backend == "nccl":
                    dist.all_reduce(view)
                else:                                   # gloo: stage through the host (testing the path without RCCL)
                    host = view.cpu()
                    dist.all_reduce(host)
                    view.copy_(host)
                h.sgd_finish(self.slot, row0, nb, view.data_ptr(), self.apply, self.w0_chunk, s, self.batch, self.flags)
                self.step_no += 1

    def synchronize(self):
        self.stream.synchronize()
        self.h.synchronize()
