"""Multi-GPU layer: independent sequences / keyframe depth updates shard across ranks (one process per GPU); the only
exchange step is the collection of finished keyframes' (idepth, idepthVar) planes on rank 0 — an RCCL gather over xGMI
(torch.distributed backend "nccl" on ROCm); the same code runs on CPU tensors with the gloo backend in tests."""
import torch
import torch.distributed as dist


class KeyframeGather:
    """Asynchronous gather of fixed-size per-keyframe records to rank 0.  One record in flight per rank: the next
    gather waits for the previous one before the send buffer is overwritten."""

    def __init__(self, shape, device, dtype=torch.float32, root=0):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.root = root
        self.send = torch.empty(shape, dtype=dtype, device=device)
        self.recv = [torch.empty(shape, dtype=dtype, device=device) for _ in range(self.world)] if self.rank == root else None
        self.pending = None
        self.collected = []      # rank 0: list of per-gather stacked records (only kept when keep=True)
        self.count = 0

    def wait(self, keep=False):
        if self.pending is not None:
            self.pending.wait()
            self.pending = None
            if self.send.is_cuda:
                # work.wait() only orders torch's current stream behind the collective; the send buffer is refilled from
                # liblsdhip's own stream, so the host has to see the gather finished before that
                torch.cuda.current_stream(self.send.device).synchronize()
            if keep and self.rank == self.root:
                self.collected.append(torch.stack([r.clone() for r in self.recv]))

    def submit(self, fill, keep=False):
        """fill(send_buffer) writes this rank's record; then the gather is started."""
        self.wait(keep)
        fill(self.send)
        self.count += 1
        if self.world == 1:
            if keep:
                self.collected.append(self.send.clone()[None])
            return
        self.pending = dist.gather(self.send, self.recv, dst=self.root, async_op=True)


def gather_keyframe_ring(ring, new, recv, root=0):
    """One collective per batch of frames: the first `new` slots of every rank's keyframe ring (slots x 2 x h x w, filled by
    lsdloop_set_keyframe_ring while the batch ran) are gathered into recv[r][:new] on `root`; recv is None elsewhere.
    All ranks must pass the same `new` (they run the same keyframe cadence).  Returns the bytes this rank contributed."""
    if new <= 0:
        return 0
    rank = dist.get_rank()
    dist.gather(ring[:new], [r[:new] for r in recv] if rank == root else None, dst=root)
    if ring.is_cuda:
        torch.cuda.current_stream(ring.device).synchronize()   # the ring is refilled from liblsdhip's stream by the next batch
    return new * ring[0].numel() * ring.element_size()
