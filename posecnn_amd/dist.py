"""Multi-GPU layer: one process per GPU, frames sharded data-parallel, ONE all-gather of the
fixed-size detection buffer per batch. The reference has no distributed code at
all (single process, `CUDA_VISIBLE_DEVICES=$1`, experiments/scripts/demo.sh:7); frames are
independent units (it loops images serially, hough_voting_gpu_op.cc:369-377 and
lib/fcn/test.py:1867), so the only exchange step is returning detections.

Backend 'nccl' is RCCL on ROCm (xGMI); 'gloo' is used for the CPU tests. The payload is
(cap+1) x 14 floats per rank (~7 KB at cap = 128): latency-bound, so it is issued as a single
`all_gather_into_tensor` with the row count packed into the last row instead of a second
collective.
"""
import os

import torch
import torch.distributed as dist

DET_COLS = 14  # box7 | pose7


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def init_from_env(backend=None, force=False):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torchrun contract).
    Returns (rank, world_size, local_rank). world_size 1 needs no process group; `force=True`
    creates one anyway (a 1-rank RCCL communicator), so that the collective path itself runs on
    a single-GPU box."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port()) if world == 1 else "29500"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def gpu_numa_node(local):
    """NUMA node of GPU `local` from sysfs (its PCI function's `numa_node`), or None when the platform does not say."""
    try:
        p = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        return node if node >= 0 else None
    except Exception:
        return None


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_gpu_numa(local):
    """One process per GPU: keep this rank's host thread — and with it, by first touch, the pinned upload slots and
    drain buffers it allocates afterwards — on the NUMA node its GPU hangs off, so that the 118 MB of H2D per step
    never cross the socket interconnect and 8 ranks do not pile onto node 0. Returns (node, n_cpus) or None if the
    topology is not exposed (containers without sysfs PCI entries) — never fatal."""
    node = gpu_numa_node(local)
    if node is None:
        return None
    try:
        cpus = _parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read())
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node, len(cpus)
    except Exception:
        return None


def time_all_gather(rows, count, iters=50, group=None):
    """Latency of the path's only collective, by itself: `iters` back-to-back all-gathers of one packed detection
    block ((cap + 1) x 14 floats per rank), microseconds each (device events on RCCL, host clock on gloo)."""
    import time
    if not dist.is_initialized():
        return None
    for _ in range(3):
        all_gather_packed(rows, count, group=group)
    if rows.is_cuda and dist.get_backend(group) != "gloo":
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            all_gather_packed(rows, count, group=group)
        e1.record()
        e1.synchronize()
        return 1000.0 * e0.elapsed_time(e1) / iters
    t0 = time.perf_counter()
    for _ in range(iters):
        all_gather_packed(rows, count, group=group)
    return 1e6 * (time.perf_counter() - t0) / iters


def shard_range(n_frames, rank, world):
    """Contiguous shard [lo, hi) of a global batch: rank r takes frames [r*N/W, (r+1)*N/W)."""
    lo = (n_frames * rank) // world
    hi = (n_frames * (rank + 1)) // world
    return lo, hi


def pack_detections(rows, count, frame_offset=0):
    """rows [cap, 14] (+ device count [1]) -> [cap+1, 14] with the count in [-1, 0] and the
    image index column shifted to global frame numbering."""
    cap = rows.shape[0]
    buf = torch.zeros((cap + 1, DET_COLS), dtype=torch.float32, device=rows.device)
    buf[:cap] = rows
    if frame_offset:
        valid = (torch.arange(cap, device=rows.device) < count).to(rows.dtype)
        buf[:cap, 0] += valid * float(frame_offset)
    buf[cap, 0] = count.to(torch.float32).reshape(())
    return buf


def all_gather_packed(rows, count, frame_offset=0, group=None, packed=None):
    """The collective itself: returns the packed buffer of every rank, [W, cap+1, 14] (count of
    rank r in [r, cap, 0]). Asynchronous with respect to the host on RCCL.

    On a `gloo` group with device tensors (the one-GPU rehearsal of the multi-rank path, `bench.py
    --backend gloo --shared-device`: RCCL refuses two ranks on one device, and gloo's all-gather takes host
    tensors only) the packed block is staged through the host: D2H on the batch's stream, wait for THAT copy,
    gather on gloo; the result is a host tensor, which `HostDrain` takes as it is."""
    # `packed`: the block already assembled on the device by ops.det_assemble(..., frame_offset=...) — same content as
    # pack_detections(rows, count, frame_offset), without its four framework launches
    buf = packed if packed is not None else pack_detections(rows, count, frame_offset)
    if not dist.is_initialized():
        return buf.unsqueeze(0)   # single process, no communicator: nothing to exchange
    world = dist.get_world_size(group)
    if buf.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(buf.shape, dtype=buf.dtype, pin_memory=True)
        host.copy_(buf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ev.synchronize()
        flat = torch.empty((world * buf.shape[0], buf.shape[1]), dtype=buf.dtype)
        dist.all_gather_into_tensor(flat, host, group=group)
        return flat.view(world, buf.shape[0], buf.shape[1])
    flat = torch.empty((world * buf.shape[0], buf.shape[1]), dtype=buf.dtype, device=buf.device)
    dist.all_gather_into_tensor(flat, buf, group=group)  # rank-major concatenation (RCCL and gloo)
    return flat.view(world, buf.shape[0], buf.shape[1])


def all_gather_detections(rows, count, frame_offset=0, group=None):
    """Every rank ends up with all ranks' detections: returns (rows [W, cap, 14], counts [W] int64).
    A no-op (plus reshape) when no process group exists."""
    gathered = all_gather_packed(rows, count, frame_offset, group)
    cap = rows.shape[0]
    return gathered[:, :cap], gathered[:, cap, 0].round().to(torch.int64)


def _flatten_host(packed):
    import numpy as np
    cap = packed.shape[1] - 1
    parts = [packed[r, :int(round(float(packed[r, cap, 0])))] for r in range(packed.shape[0])]
    return np.concatenate(parts) if parts else np.zeros((0, DET_COLS), np.float32)


def flatten_gathered(rows, counts):
    """[W, cap, 14] + counts -> [sum(counts), 14] in rank (= global frame) order, on the host."""
    rows = rows.cpu().numpy()
    counts = counts.cpu().numpy()
    import numpy as np
    parts = [rows[r, :int(counts[r])] for r in range(rows.shape[0])]
    return np.concatenate(parts) if parts else np.zeros((0, DET_COLS), np.float32)


class HostDrain:
    """Device -> host drain of the gathered detection buffer without stalling the launch thread.

    `submit` enqueues an asynchronous copy of the packed [W, cap+1, 14] buffer into one of `depth`
    pinned host buffers and records an event; `collect` waits for THAT event only and returns the
    flattened [sum(counts), 14] rows. With depth 2 the host can launch batch i+1 before it collects
    batch i, so the D2H latency and the host-side NMS overlap the next batch's kernels (on CPU
    tensors the copy is synchronous and the event is skipped)."""

    def __init__(self, depth=2, spin=False):
        # spin: poll the copy's event instead of sleeping on it — a latency-critical single-frame loop (bench.py --latency)
        # gets its detections a wake-up (tens of microseconds) earlier at the price of a busy host core
        self.spin = spin
        self.depth = depth
        self.slots = [None] * depth
        self.events = [None] * depth
        self.busy = [False] * depth
        self.n = 0

    def submit(self, packed):
        k = self.n % self.depth
        if self.busy[k]:
            raise RuntimeError("HostDrain: %d batches in flight, collect() one before submitting more" % self.depth)
        self.n += 1
        self.busy[k] = True
        on_gpu = packed.is_cuda
        if self.slots[k] is None or self.slots[k].shape != packed.shape:
            self.slots[k] = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=on_gpu)
            self.events[k] = torch.cuda.Event() if on_gpu else None
        self.slots[k].copy_(packed, non_blocking=True)
        if on_gpu:
            self.events[k].record()
        return k

    def collect(self, ticket):
        if not self.busy[ticket]:
            raise RuntimeError("HostDrain: ticket %d was already collected" % ticket)
        if self.events[ticket] is not None:
            if self.spin:
                while not self.events[ticket].query():
                    pass
            else:
                self.events[ticket].synchronize()
        out = _flatten_host(self.slots[ticket].numpy())  # concatenation copies out of the pinned slot
        self.busy[ticket] = False
        return out


def barrier():
    if dist.is_initialized():
        dist.barrier()


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def gather_over_ranks(value, device):
    """Every rank's python float, in rank order (bench.py's per-rank step times: the MAX alone cannot say whether one
    rank or all of them were slow). A collective: every rank calls it."""
    if not dist.is_initialized():
        return [float(value)]
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = torch.empty(dist.get_world_size(), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, t)
    return [float(v) for v in out.cpu().tolist()]


def max_over_ranks(value, device):
    """MAX all-reduce of a python float (timing contract of bench.py)."""
    if not dist.is_initialized():
        return float(value)
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
