"""One-process-per-GPU plumbing for the two drivers.

Mirror of the slice of the reference's utils.py the hot path uses
(utils.py:214-281: ``init_distributed_mode``, rank helpers, print muting), with two
deliberate differences:
  * work is split in BALANCED contiguous blocks (the reference's ``len//world + 1``
    steps leave ranks empty and crash them, run_visual_tokenization.py:265,427-431);
    contiguity is kept so rank-order concatenation reproduces single-process order;
  * per-rank results are gathered to rank 0 as UTF-8 JSON bytes over the process
    group (RCCL over xGMI with backend 'nccl', Gloo on CPU) instead of tmp files on a
    shared filesystem + barrier (run_video_CapFilt.py:249-291).  This is the only
    collective on the path: the videos shard with no data-path exchange.
"""
from __future__ import annotations

import datetime
import json
import os

import torch
import torch.distributed as dist


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def setup_for_distributed(is_master):
    """utils.py:214-226: mute print on non-master ranks (force=True overrides)."""
    import builtins

    builtin_print = builtins.print

    def print(*args, **kwargs):  # noqa: A001
        force = kwargs.pop("force", False)
        if is_master or force:
            builtin_print(*args, **kwargs)

    builtins.print = print


def init_distributed_mode(args=None, backend=None, mute=False):
    """env:// rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns (rank, world, local_rank)."""
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        if args is not None:
            args.distributed = False
        return 0, 1, 0
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank,
                                timeout=datetime.timedelta(seconds=7200))
    barrier()
    if args is not None:
        args.rank, args.world_size, args.gpu, args.distributed = rank, world, local, True
    if mute:
        setup_for_distributed(rank == 0)
    return rank, world, local


def barrier():
    if is_dist_avail_and_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def shard_bounds(n, world=None, rank=None):
    """Balanced contiguous block [start, end) of ``n`` items for ``rank``."""
    world = get_world_size() if world is None else world
    rank = get_rank() if rank is None else rank
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _comm_device():
    if is_dist_avail_and_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


#: backends whose ProcessGroup implements ``gather`` (torch 2.x: ProcessGroupNCCL — RCCL on ROCm — and ProcessGroupGloo both do;
#: ucc / mpi builds are not exercised here and take the all_gather form)
_GATHER_BACKENDS = ("nccl", "gloo")


def _gather_supported():
    """Same answer on every rank by construction: the environment switch and the backend's name, nothing measured."""
    if os.environ.get("VIDIL_GATHER", "gather") == "allgather":
        return False
    return str(dist.get_backend()).lower() in _GATHER_BACKENDS


def ranks_seen():
    """How many DISTINCT devices the job's ranks sit on: an ``all_gather`` of a 16-byte device identity (the GPU's UUID; host
    name hash + pid for a CPU rank), counted on every rank.  An N-GPU run proves "one rank per GPU, N different GPUs" in its
    own output with it (bench.py: ``config.ranks_seen``)."""
    import hashlib
    import socket

    ident = torch.zeros(16, dtype=torch.uint8)
    if torch.cuda.is_available() and _comm_device().type == "cuda":
        idx = torch.cuda.current_device()
        props = torch.cuda.get_device_properties(idx)
        uuid = getattr(props, "uuid", None)
        if uuid is not None and hasattr(uuid, "bytes"):
            raw = bytes(uuid.bytes)[:16]
        else:   # (no UUID from this torch build: host + PCI bus id / ordinal still keeps the GPUs of a job apart)
            raw = hashlib.sha1(f"{socket.gethostname()}:{getattr(props, 'pci_bus_id', idx)}:{idx}".encode()).digest()[:16]
    else:
        raw = hashlib.sha1(f"{socket.gethostname()}:{os.getpid()}".encode()).digest()[:16]
    ident[:len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    if not is_dist_avail_and_initialized():
        return 1
    dev = _comm_device()
    mine = ident.to(dev)
    every = [torch.zeros(16, dtype=torch.uint8, device=dev) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    return len({bytes(t.cpu().numpy().tobytes()) for t in every})


def gather_json(obj):
    """Gather one JSON-serialisable object per rank (rank order); rank 0 gets the list, other ranks get None.

    Two collectives and nothing else: an ``all_gather`` of the payload sizes (one int64 per rank), then ONE ``gather`` TO RANK 0
    of the UTF-8 bytes padded to the longest payload (run_video_CapFilt.py:261-291 / run_visual_tokenization.py:447-463: only
    rank 0 merges and writes).  RCCL (backend 'nccl') moves device buffers over xGMI, Gloo moves host buffers.  The same code
    path runs for every world size including 1, so what an 8-GPU job executes is what the one-GPU box has already executed
    (tests/test_dist_gpu.py).  A true gather since round 5; ``all_gather`` of the payloads remains for a backend known not to
    implement ``gather`` and behind $VIDIL_GATHER=allgather (``_gather_supported``: decided before the collective)."""
    if not is_dist_avail_and_initialized():
        return [obj]
    dev = _comm_device()
    raw = json.dumps(obj).encode("utf-8")
    world, rank = dist.get_world_size(), dist.get_rank()
    size = torch.tensor([len(raw)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    mine = torch.zeros(cap, dtype=torch.uint8)
    mine[:len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    mine = mine.to(dev)
    # the collective is chosen BEFORE any is issued, from facts every rank shares (the env switch and the backend's name):
    # nothing is decided by catching an exception around a collective — an error raised on some ranks only (an OOM on rank 0)
    # would otherwise send those to another collective than the rest of the job sits in.  Real errors propagate.
    use_gather = _gather_supported()
    if use_gather:
        bufs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, gather_list=bufs, dst=0)
    else:
        bufs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(bufs, mine)
    if rank != 0:
        return None
    return [json.loads(bufs[r][:sizes[r]].cpu().numpy().tobytes().decode("utf-8")) for r in range(world)]


def merge_rank_dicts(parts):
    """dict.update in rank order, as run_video_CapFilt.py:272-283 / run_visual_tokenization.py:454-457."""
    out = {}
    for p in parts:
        out.update(p)
    return out


def max_over_ranks(value: float) -> float:
    if not is_dist_avail_and_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
