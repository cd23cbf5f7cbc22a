"""Data parallelism for the STrajNet train step: one process per GPU, torch.distributed over RCCL (backend "nccl"
on ROCm) / xGMI.  Mirrors the reference's tf.distribute.MirroredStrategy semantics (train.py:69,158-197,295,319):
  * scenes are independent -> contiguous batch split B/G per rank (`shard_batch`), weights replicated;
  * the per-replica loss is pre-divided by the replica count (OGMFlow_loss(replica=G), loss.py:200,229,250,294) and the
    gradients are SUM-all-reduced => global-batch mean;
  * ONE exchange step per iteration: the model's flat f32 gradient buffer (13.28 M values, 53 MB) is the bucket --
    a single all-reduce, no per-tensor collectives (xGMI is point-to-point: few large messages).
  * `OverlappedGradSync`: the same sum as TWO buckets of that buffer.  Backward produces the gradients of everything downstream
    of the raster encoder first (decoder, cross-attentions, trajNet, FG-MSA: the tail of the flat buffer, ~8 M values); their
    all-reduce is launched on RCCL's stream as soon as they exist and runs under the encoder's backward; the encoder bucket
    (~5.4 M values) follows.  Needs the model in cut_encoder mode (modules.STrajNet.backward_encoder).
Forward-only needs no communication (replicas only).
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_batch(batch, rank_=None, world_=None):
    """Contiguous split of every [B, ...] tensor of `batch` (dict) across ranks (experimental_distribute_dataset)."""
    r = rank() if rank_ is None else rank_
    w = world() if world_ is None else world_
    out = {}
    for k, v in batch.items():
        B = v.shape[0]
        if B % w:
            raise ValueError(f'global batch {B} of {k} not divisible by world size {w}')
        per = B // w
        out[k] = v[r * per:(r + 1) * per]
    return out


def allreduce_flat_grads(flat_grads, async_op=False):
    """SUM all-reduce of the flat gradient bucket (RCCL on GPU tensors, gloo on CPU tensors)."""
    if world() == 1:
        return None
    return dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, async_op=async_op)


def broadcast_weights(flat_weights, src=0):
    """Make replicas bit-identical at start (MirroredStrategy creates mirrored variables from one initial value)."""
    if world() > 1:
        dist.broadcast(flat_weights, src=src)


def reduce_metrics(values):
    """Mean over replicas of a small f32 tensor of scalars (train.py:159-170,226-229)."""
    if world() > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
        values /= world()
    return values


class OverlappedGradSync:
    """Two-bucket gradient exchange overlapped with backward (the reference's all-reduce is part of the step: train.py:224).

        model.cut_encoder = True
        loss.backward()              # ends at the encoder outputs: flat_grads()[split:] is final
        sync.tail()                  # async SUM all-reduce of that bucket (RCCL stream), returns at once
        model.backward_encoder()     # runs concurrently with it
        sync.head_and_wait()         # all-reduce of flat_grads()[:split]; the current stream then waits for both

    With a captured step the two halves are two hipGraphs (graph.GraphedTrainStep(split=True)) and tail() is its `between` hook."""

    def __init__(self, model):
        g = model.flat_grads()
        k = int(model.bucket_split)
        self.head, self.tail_bucket = g[:k], g[k:]
        self._work = []

    def tail(self):
        if world() > 1:
            self._work.append(dist.all_reduce(self.tail_bucket, op=dist.ReduceOp.SUM, async_op=True))

    def head_and_wait(self):
        if world() > 1:
            self._work.append(dist.all_reduce(self.head, op=dist.ReduceOp.SUM, async_op=True))
            for w in self._work:
                w.wait()             # stream-ordered for RCCL (no host block); blocking for gloo
        self._work = []

    @staticmethod
    def info():
        """What the communicator looks like from this rank (printed by bench.py at N>1 so a scaling run can be audited)."""
        d = {'backend': dist.get_backend() if world() > 1 else None, 'ranks': world()}
        try:
            v = torch.cuda.nccl.version()
            d['rccl_version'] = '.'.join(str(x) for x in v) if isinstance(v, tuple) else str(v)
        except Exception:
            d['rccl_version'] = None
        return d
