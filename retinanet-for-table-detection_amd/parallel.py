"""Data parallelism for the training step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

What the reference's multi_gpu_model defines and this preserves (RetinaNet.py:105-116):
  * identical weights on every replica, the minibatch sliced on axis 0;
  * ONE loss over the merged batch: focal and smooth-L1 are normalised by the positive-anchor count of the WHOLE batch
    (model/losses.py:39-44,87-90) -> the per-rank [sum, sum, count, count] vector is sum-all-reduced before the loss
    backward (4 doubles), so the per-rank gradients simply add up to the merged-batch gradient;
  * gradients: sum-all-reduce of the flat f32 gradient buffer (145 MB for R50), in buckets launched as soon as the last
    weight gradient of a bucket has been enqueued (backward produces layers in reverse order: heads -> FPN -> C5..C2), on
    a side stream so the transfers hide under the remaining backward kernels; the global-norm clip then needs no extra
    collective because every rank holds the full reduced gradient.
Inference shards images with no collective at all.
"""
import torch
import torch.distributed as dist


def allreduce_loss_sums(loss_sums, group=None):
    """[sum focal, sum smooth-L1, #pos (labels), #pos (regression)] of this rank -> of the merged batch (new tensor)."""
    out = loss_sums.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out


class GradBucketer:
    """Bucketed, overlapped sum-all-reduce of a flat gradient buffer.

    segments: [(name, start, end)] in FORWARD order (element offsets into `flat`).  Buckets are contiguous runs of segments
    cut from the END of the buffer (the order backward finishes them) at about `bucket_bytes`.  layer_done(name) is called
    right after the kernels producing that segment were enqueued, ON THE STREAM they were enqueued on (the trainer spreads the
    weight gradients over several HIP streams): it records an event there, one per (bucket, stream) - streams are in order, so the
    latest event of a stream covers every earlier layer on it.  When a bucket is complete its all-reduce is issued on the
    communication stream behind ALL of the bucket's events.  finish() issues what is left and makes the compute stream wait."""

    def __init__(self, flat, segments, group=None, bucket_bytes=32 << 20, stream_layer=None):
        """stream_layer: an object with torch.cuda's current_stream / Stream / Event / stream (tests pass a recording stub and drive the
        event logic on the CPU); None = torch.cuda when `flat` lives on a GPU, no streams at all otherwise."""
        self.flat, self.group = flat, group
        self.tc = stream_layer if stream_layer is not None else torch.cuda
        self.cuda = flat.is_cuda or stream_layer is not None
        self.comm = self.tc.Stream(device=flat.device) if self.cuda else None
        es = flat.element_size()
        self.buckets, cur, size = [], [], 0
        for seg in reversed(segments):
            cur.append(seg)
            size += (seg[2] - seg[1]) * es
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.where = {seg[0]: bi for bi, b in enumerate(self.buckets) for seg in b}
        self._events = {}
        self.reset()

    def reset(self):
        self.pending = [set(seg[0] for seg in b) for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.work = []
        self.last_marks = [set(m) for m in getattr(self, "marks", [])]     # streams each bucket of the step before waited for (tests)
        self.marks = [{} for _ in self.buckets]          # bucket -> {stream id: (stream, event recorded after its last layer there)}

    def _launch(self, bi):
        segs = self.buckets[bi]
        lo, hi = min(s[1] for s in segs), max(s[2] for s in segs)
        view = self.flat[lo:hi]
        self.launched[bi] = True
        if self.cuda:
            marks = self.marks[bi]
            if not marks:                                 # finish() on a bucket no layer reported: order behind the caller's stream
                self._mark(bi)
            with self.tc.stream(self.comm):
                for _, ev in marks.values():
                    self.comm.wait_event(ev)
                self.work.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.work.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _mark(self, bi):
        st = self.tc.current_stream(self.flat.device)
        pool = self._events.setdefault((bi, st.cuda_stream), self.tc.Event())
        pool.record(st)
        self.marks[bi][st.cuda_stream] = (st, pool)

    def mark(self, name):
        """Record the current stream's progress for `name`'s bucket without completing the segment (a segment produced by
        kernels on several streams - the fused bias gradients - is marked once per stream before layer_done)."""
        bi = self.where.get(name)
        if bi is not None and not self.launched[bi] and self.cuda:
            self._mark(bi)

    def layer_done(self, name):
        bi = self.where.get(name)
        if bi is None or self.launched[bi]:
            return
        if self.cuda:
            self._mark(bi)
        self.pending[bi].discard(name)
        if not self.pending[bi]:
            self._launch(bi)

    def finish(self):
        for bi in range(len(self.buckets)):
            if not self.launched[bi]:
                self._launch(bi)
        for w in self.work:
            w.wait()
        if self.cuda:
            self.tc.current_stream(self.flat.device).wait_stream(self.comm)
        self.reset()
