"""Data parallelism over batch rows: one process per GPU, torch.distributed ("nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for the tests).

The path shards naturally over batch rows: RSSM rows, conv frames and
imagined rows are independent, every loss is a mean over rows.  Each rank therefore runs the
whole update on its rows with losses scaled by 1/GLOBAL rows (sum-of-sums, exact for uneven
shards), and there is exactly one exchange step per optimiser: a SUM all-reduce of the flat
gradient buffer before the global-norm clip, plus the scalar KL sum that drives the dual variable.  The
model gradient (5.17 M floats = 20.7 MB) goes out as TWO buckets in the order the backward finishes them:
decoder + reward head (15.7 MB, final when the decoder backward joins) while the encoder backward still
runs, then encoder + RSSM; actor and critic gradients share one buffer and are ONE 1.18 MB bucket.  Parameters, Adam moments and
log_beta are replicated and stay bit-identical because every rank applies the same update to the
same reduced gradient.  The reference has no distributed code; this is new.

xGMI is point-to-point (7 links per GPU): buckets are few and large (15.7 + 5.0 + 1.2 MB) so RCCL can
use all links at once instead of serialising many small rings.
"""
from fractions import Fraction

import torch
import torch.distributed as dist


def shard_rows(n_rows, world_size, rank):
    """Contiguous [start, stop) of `rank` when n_rows are dealt as evenly as possible
    (50 rows over 8 ranks -> 7,7,6,6,6,6,6,6)."""
    base, extra = divmod(n_rows, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class DataParallel:
    def __init__(self, group=None):
        self.group = group if group is not None else dist.group.WORLD
        self.world_size = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        self._ratio = None        # global rows / this rank's rows (exact), set by the first global_count()
        self.shard_counts = None

    # ---- collectives -------------------------------------------------------------------
    def all_reduce(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_prefix(self, buf, n):
        dist.all_reduce(buf[:n], op=dist.ReduceOp.SUM, group=self.group)
        return buf

    def all_reduce_status(self, word):
        """MAX all-reduce of an update's int32 status word (ops.take_scan_status): bit patterns are small non-negative
        integers (REPO_SCAN_STATUS_*: 1 = forward, 2 = reverse, 3 = both), so the maximum is non-zero on every rank as
        soon as it is on one -- and every rank then skips the same optimiser steps and raises in the same update
        instead of one rank raising alone while its peers wait in the next gradient all-reduce."""
        dist.all_reduce(word, op=dist.ReduceOp.MAX, group=self.group)
        return word

    def all_reduce_begin(self, t, stream=None):
        """Start a SUM all-reduce of `t` (a contiguous slice of a flat gradient buffer) so that the kernels
        issued next overlap it (the encoder backward while the decoder's 15.7 MB bucket is on the wire);
        `all_reduce_end` orders the current stream behind it.  Every rank must begin its buckets in the same order.

        stream: an EXISTING idle stream of the caller to issue the collective on (it first waits for the current
        stream).  A blocking collective runs on the stream it is called from, so no new stream appears: HIP
        multiplexes all streams onto 4 hardware queues and a further concurrent stream (the communicator's
        own, which async_op=True would use) cost the update more than the overlap returned (measured with
        one rank: 10.4 vs 9.5 ms per update).  Without `stream` (CPU / gloo) the work handle of an
        asynchronous collective is returned."""
        if stream is None:
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        stream.wait_stream(torch.cuda.current_stream(t.device))
        with torch.cuda.stream(stream):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return stream

    def all_reduce_end(self, works):
        for w in works:
            if isinstance(w, torch.cuda.Stream):
                torch.cuda.current_stream(w.device).wait_stream(w)
            else:
                w.wait()

    def broadcast(self, t, src=0):
        dist.broadcast(t, src=src, group=self.group)
        return t

    def barrier(self):
        dist.barrier(group=self.group)

    def max_float(self, x, device=None):
        t = torch.tensor([float(x)], dtype=torch.float64, device=device or self._dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def global_count(self, local_count):
        """Sum of `local_count` over ranks, with NO communication in the steady state.

        Every count the update asks about (T*B rows, the N = T*B imagination starts) is a fixed multiple
        of the rank's batch shard, so the first call all-gathers the ranks' counts ONCE and keeps the exact
        ratio global/local; later calls scale by it.  This is collective-free on every rank alike -- a
        per-value cache would issue its all-reduce only on the ranks that miss, and hang the others.
        Shard sizes are therefore fixed for the lifetime of this object; call `reset_counts()` on ALL
        ranks together before changing them."""
        local_count = int(local_count)
        if self._ratio is None:
            mine = torch.tensor([local_count], dtype=torch.int64, device=self._dev)
            everyone = [torch.zeros_like(mine) for _ in range(self.world_size)]
            dist.all_gather(everyone, mine, group=self.group)
            counts = [int(t.item()) for t in everyone]
            # validated on the GATHERED list, identically on every rank: an empty shard anywhere makes ALL ranks
            # raise together (a rank raising alone would leave its peers blocked in the next gradient all-reduce)
            if any(n <= 0 for n in counts) or counts[self.rank] != local_count:
                raise RuntimeError(f"global_count: bad shard sizes {counts} (rank {self.rank} has {local_count})")
            self._ratio = Fraction(sum(counts), local_count)
            self.shard_counts = counts
        total = self._ratio * local_count
        if total.denominator != 1:
            # a per-rank condition (only this rank knows its argument): the caller must abort the whole group,
            # e.g. let the exception end the process so that the launcher tears the job down
            raise RuntimeError(f"global_count({local_count}): not a multiple of this rank's shard "
                               f"(global/local = {self._ratio}); call reset_counts() on all ranks first")
        return int(total)

    def reset_counts(self):
        """Collective by convention: every rank forgets the shard ratio; the next global_count() re-gathers."""
        self._ratio = None
        self.shard_counts = None

    _dev = torch.device("cpu")

    # ---- agent hookup ------------------------------------------------------------------
    def attach(self, agent):
        """Make the replicas identical (rank 0's parameters and optimiser state win) and route
        the agent's gradient / scalar exchanges through this group."""
        self._dev = agent.device
        for opt in (agent.model_optimizer, agent.actor_optimizer, agent.value_optimizer):
            for buf in (opt.flat, opt.exp_avg, opt.exp_avg_sq):
                self.broadcast(buf)
        if hasattr(agent, "log_beta"):
            self.broadcast(agent.log_beta)
            self.broadcast(agent.beta_optimizer.exp_avg)
            self.broadcast(agent.beta_optimizer.exp_avg_sq)
        # every rank draws its OWN reparameterisation noise for its rows: decorrelate the in-kernel Philox streams
        if hasattr(agent, "_noise_seed"):
            agent._noise_seed = (agent._noise_seed + self.rank * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)
        agent.dp = self
        return agent
