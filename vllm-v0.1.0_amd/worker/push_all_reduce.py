"""One-shot all-reduce over peer-mapped memory, for the tensor-parallel group of one node.

The reference sums the row-parallel partial results with NCCL's all-reduce
(vllm/model_executor/parallel_utils/parallel_state.py:549-556; call site
parallel_utils/tensor_parallel/layers.py:416-426).  For the message sizes of that call
site (KiB to a few MiB) on MI355X, where every GPU pair has an xGMI link of its own,
`PushAllReduce` is the alternative collective: every rank writes its buffer into an inbox on
every peer, all links at once, and sums the inboxes locally — one kernel launch, one hop
(csrc/pa_comm.hip).  An instance is a `reduce_fn` for `GraphAllReduce`: the launcher keeps
its side stream, events and HIP graphs, only the collective inside changes.

    reduce = PushAllReduce(max_num_tokens * hidden_size * element_size, group)
    launcher = GraphAllReduce(max_num_tokens, hidden_size, dtype, group=group, reduce_fn=reduce)

Set-up exchanges IPC handles over the group (one process per GPU) — any backend that can
all_gather Python objects will do — and ends with a barrier, as the first reduce needs every
rank connected.  `PushAllReduce.local_ranks` builds the ranks of a group inside one process
(one GPU with a stream per rank, or one process driving several GPUs with peer access).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from .. import _capi


class PushAllReduce:

    MAX_RANKS = 8

    AUTO, ONE_SHOT, TWO_SHOT = _capi.COMM_AUTO, _capi.COMM_ONE_SHOT, _capi.COMM_TWO_SHOT

    def __init__(self, max_bytes: int, group: Optional[dist.ProcessGroup] = None,
                 device: Optional[torch.device] = None, algorithm: int = _capi.COMM_AUTO, _local=None) -> None:
        """algorithm: AUTO (by message size and group size), ONE_SHOT (the whole message to every peer,
        one round of flags) or TWO_SHOT (each rank sums one N-th of it: 2 / N of the bytes per link, two
        rounds of flags); the same on every rank."""
        self.max_bytes = int(max_bytes)
        self.algorithm = int(algorithm)
        self._comm = None
        if _local is not None:  # (rank, world): connected later by local_ranks
            self.rank, self.world = _local
        else:
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > self.MAX_RANKS:
            raise ValueError(f"{self.world} ranks: the one-shot all-reduce is for at most {self.MAX_RANKS} GPUs of a node")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._group = group
        if _local is not None:
            with torch.cuda.device(self.device):
                self._comm, self._handles = _capi.comm_create(self.rank, self.world, self.max_bytes)
            return
        # Set-up succeeds or fails on ALL ranks: a rank that raised alone would leave the others
        # waiting in the next collective of the group.
        error: Optional[str] = None
        handles = b""
        try:
            with torch.cuda.device(self.device):
                self._comm, handles = _capi.comm_create(self.rank, self.world, self.max_bytes)
        except Exception as e:  # noqa: BLE001
            error = f"{type(e).__name__}: {e}"
        gathered: List[Optional[bytes]] = [None] * self.world
        dist.all_gather_object(gathered, None if error else handles, group=group)
        if error is None and all(h is not None for h in gathered):
            try:
                with torch.cuda.device(self.device):
                    _capi.comm_connect(self._comm, gathered)
            except Exception as e:  # noqa: BLE001
                error = f"{type(e).__name__}: {e}"
        elif error is None:
            error = "another rank could not create its communicator"
        self._agree(error, "set-up")  # also the barrier: no rank reduces before every rank has mapped every inbox

    def _agree(self, error: Optional[str], what: str) -> None:
        """Collective: raises on every rank if any rank reports an error."""
        errors: List[Optional[str]] = [None] * self.world
        dist.all_gather_object(errors, error, group=self._group)
        failed = [(r, e) for r, e in enumerate(errors) if e]
        if failed:
            self.close()
            raise RuntimeError(f"one-shot all-reduce {what} failed on rank {failed[0][0]}: {failed[0][1]}")

    def self_test(self) -> None:
        """Collective: one small reduce of known values on every rank's current stream, checked on every
        rank (sum, no wait given up); raises on every rank if any rank saw a wrong result.  A few seconds at
        most even when peers cannot be reached (the kernel's waits are bounded)."""
        error: Optional[str] = None
        try:
            n = min(4096, self.max_bytes // 2 // 8 * 8)
            x = torch.full((n,), float(self.rank + 1), dtype=torch.float16, device=self.device)
            self(x)
            torch.cuda.synchronize(self.device)
            want = float(self.world * (self.world + 1) // 2)
            gave_up = self.status()[1]
            if gave_up or not bool((x == want).all()):
                error = f"sum {float(x[0])} (expected {want}), waits given up {gave_up}"
        except Exception as e:  # noqa: BLE001
            error = f"{type(e).__name__}: {e}"
        self._agree(error, "self-test")

    @classmethod
    def local_ranks(cls, world: int, max_bytes: int, devices: Optional[Sequence[torch.device]] = None,
                    algorithm: int = _capi.COMM_AUTO) -> List["PushAllReduce"]:
        """All ranks of a group inside this process, connected by pointer.  devices: one per rank
        (default: the current device for every rank)."""
        devices = list(devices) if devices is not None else [torch.device("cuda", torch.cuda.current_device())] * world
        ranks = [cls(max_bytes, device=devices[r], algorithm=algorithm, _local=(r, world)) for r in range(world)]
        for r in ranks:
            with torch.cuda.device(r.device):
                _capi.comm_connect_local(r._comm, [x._comm for x in ranks])
        return ranks

    def __call__(self, t: torch.Tensor) -> None:
        """In-place sum of t over the ranks, queued on the current stream of t's device."""
        if not t.is_contiguous():
            raise ValueError("the one-shot all-reduce takes a contiguous tensor")
        if t.device != self.device:
            raise ValueError(f"tensor on {t.device}, communicator on {self.device}")
        _capi.comm_all_reduce(self._comm, t, torch.cuda.current_stream(self.device), self.algorithm)

    def status(self):
        """(reduces completed, waits given up); synchronises with the device.  A reduce whose wait was
        given up (a peer did not arrive within two seconds) returns NaNs, and so does every later one
        on this object: replace it."""
        with torch.cuda.device(self.device):
            return _capi.comm_status(self._comm)

    def close(self) -> None:
        if self._comm is not None:
            with torch.cuda.device(self.device):
                torch.cuda.synchronize(self.device)
                _capi.comm_destroy(self._comm)
            self._comm = None
