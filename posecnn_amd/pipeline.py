"""Host-side frame pipeline of the batched inference driver (the MI355X counterpart of the
reference's per-frame loop `for i in perm: im_segment_single_frame(...)`, lib/fcn/test.py:1867-1945,
which builds its blobs on the host and hands them to `sess.run` through `feed_dict`, :151-195).

  FrameUploader  the feed_dict hand-over: image blobs wait in pinned host memory and travel to
                 HBM on a side HIP stream, `depth` batches ahead of the kernels that consume them,
                 so the PCIe copy (59 MB per 16 COLOR frames, 118 MB RGB-D) overlaps the previous
                 batch's backbone instead of sitting in front of it.
  GraphedStep    (optional) one whole batch captured in a hipGraph and replayed: the ~600 kernel
                 launches of a step become one host call (lib/fcn/test.py has no equivalent; TF1's
                 executor plays that role there).
"""
import torch


class FrameUploader:
    """Pinned host blobs -> device slots on a side stream.

    `host_batches` is a list of tuples of pinned CPU tensors (one tuple per distinct batch; the
    bench cycles through them). `get(i)` returns the device tensors of batch i and makes the
    CURRENT stream wait for their copy; it also enqueues the copy of batch i + depth. A slot is
    re-used only after the compute stream has passed the point where `release(i)` was called for
    the batch that last lived in it."""

    def __init__(self, host_batches, device, depth=2):
        assert depth >= 1
        self.host = host_batches
        self.device = torch.device(device)
        self.depth = depth
        self.nslots = depth + 1
        self.stream = torch.cuda.Stream(device=self.device)
        self.slots = [None] * self.nslots
        self.ready = [torch.cuda.Event() for _ in range(self.nslots)]
        self.free = [None] * self.nslots       # event recorded on the compute stream by release()
        self.loaded = [-1] * self.nslots
        self.next_to_issue = 0
        self.bytes_per_batch = sum(t.numel() * t.element_size() for t in host_batches[0] if t is not None)

    def _issue(self, i):
        k = i % self.nslots
        src = self.host[i % len(self.host)]
        if self.slots[k] is None:
            self.slots[k] = alloc_adjacent(src, self.device)
        with torch.cuda.stream(self.stream):
            if self.free[k] is not None:
                self.stream.wait_event(self.free[k])   # the kernels that read the old contents are done
            for d, h in zip(self.slots[k], src):
                if d is not None:
                    d.copy_(h, non_blocking=True)
            self.ready[k].record(self.stream)
        self.loaded[k] = i

    def get(self, i):
        """Device tensors of batch i (call with i = 0, 1, 2, ... in order)."""
        while self.next_to_issue <= i + self.depth - 1:
            self._issue(self.next_to_issue)
            self.next_to_issue += 1
        k = i % self.nslots
        assert self.loaded[k] == i, "FrameUploader.get() must be called in batch order"
        torch.cuda.current_stream(self.device).wait_event(self.ready[k])
        return self.slots[k]

    def release(self, i):
        """The compute stream has enqueued every kernel that reads batch i."""
        k = i % self.nslots
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.free[k] = ev


def alloc_adjacent(like, device):
    """Device tensors shaped like the (CPU) tensors in `like`, carved in order out of ONE allocation
    (256-byte aligned). Blobs of equal shape that follow each other — the colour and the depth blob of
    an RGB-D batch — are then adjacent in HBM, and `stacked_view` can hand both towers to one grouped
    launch without a concatenation pass."""
    offs, total = [], 0
    for t in like:
        offs.append(total)
        if t is not None:
            total += (t.numel() * t.element_size() + 255) // 256 * 256
    flat = torch.empty(max(total, 1), dtype=torch.uint8, device=device)
    return tuple(None if t is None else flat[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
                 for t, o in zip(like, offs))


def stacked_view(a, b):
    """[2B, ...] view over `a` followed by `b` if they are adjacent contiguous blocks of one allocation
    (see alloc_adjacent), else None."""
    if (a.shape != b.shape or a.dtype != b.dtype or not a.is_contiguous() or not b.is_contiguous()
            or a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()
            or b.data_ptr() != a.data_ptr() + a.numel() * a.element_size()):
        return None
    return a.as_strided((2 * a.shape[0],) + tuple(a.shape[1:]), a.stride())


def pin(t):
    """Page-locked copy of a CPU tensor (what the feed_dict blobs live in)."""
    return t.contiguous().pin_memory() if t is not None else None


class GraphedStep:
    """One whole batch step captured in a hipGraph (torch.cuda.CUDAGraph on ROCm) and replayed:
    the ~150 kernel launches of a step (gfx950 library kernels + the few framework ops left)
    become ONE host call, so a batch-1 frame is no longer bound by the host's launch rate.

    `fn()` must enqueue the step on the current stream with static shapes, read its inputs from
    fixed tensors (the caller refills them before `replay()`), never synchronise with the host,
    and return a tensor / tuple of tensors — those live in the graph's private pool and are
    overwritten by the next replay. Every entry of libposecnn_hip.so is capture-legal (no
    allocation, no sync, stream argument); kernel timing (`pcnn_profile_enable`) must be off."""

    def __init__(self, fn, warmup=3, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # The step is warmed up AND captured on a stream of its own, kept alive with the graph: the library's scratch
        # (Hough / ADL workspaces, split-K and Cin-split partials, ticket counters) is keyed by the stream that is current
        # when it is requested (ops._ws), so every GraphedStep bakes in ITS OWN buffers and two of them can replay
        # concurrently on two streams (bench.py --graph --streams 2). Captured on torch's shared capture stream, two
        # graphs would share one set of scratch buffers — a race when they overlap.
        self.side = torch.cuda.Stream(device=self.device)
        self.side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):     # warm-up off the default stream: MIOpen find, workspaces, caches
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream(self.device).wait_stream(self.side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.side):
            self.outputs = fn()
        torch.cuda.synchronize(self.device)

    def replay(self):
        self.graph.replay()
        return self.outputs
