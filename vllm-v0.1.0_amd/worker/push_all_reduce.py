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
rank connected, and then with `self_test()`: both algorithms on a small and on a multi-MiB message,
checked on every rank, because the kernel's ordering rests on a property of the link that only a
run on the link can show (a peer's acknowledgement of a write-through store means the data is
visible there before the flag that follows it).  Serving code must not skip it (`self_test=False`
is for tests that build a broken group on purpose).  `PushAllReduce.local_ranks` builds the ranks
of a group inside one process (one GPU with a stream per rank, or one process driving several
GPUs with peer access).

The ranks' processes map each other's device memory through HIP IPC handles.  On hosts whose
driver only offers dmabuf IPC, `HSA_ENABLE_IPC_MODE_LEGACY=0` must be in the environment of every
rank before the HIP runtime starts; without it `hipIpcGetMemHandle` fails with "invalid argument"
and set-up raises on every rank, naming the variable.

A reduce waits for its peers for at most `wait_seconds` (a bound, so that a dead peer cannot hang
the GPU); a wait that is given up leaves NaNs in that rank's result and breaks the communicator.
`check()` — called by `GraphAllReduce.check()` — reads the counters back and raises: call it where
the step synchronises anyway, before the results are used.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from .. import _capi


class PushAllReduce:

    MAX_RANKS = 8

    AUTO, ONE_SHOT, TWO_SHOT = _capi.COMM_AUTO, _capi.COMM_ONE_SHOT, _capi.COMM_TWO_SHOT

    DEFAULT_WAIT_SECONDS = 30.0

    def __init__(self, max_bytes: int, group: Optional[dist.ProcessGroup] = None,
                 device: Optional[torch.device] = None, algorithm: int = _capi.COMM_AUTO,
                 wait_seconds: Optional[float] = None, self_test: bool = True, _local=None) -> None:
        """algorithm: AUTO (by message size and group size), ONE_SHOT (the whole message to every peer,
        one round of flags) or TWO_SHOT (each rank sums one N-th of it: 2 / N of the bytes per link, two
        rounds of flags); the same on every rank.
        wait_seconds: how long a reduce waits for its peers (default 30 s: longer than any stall a
        healthy rank has between two reduces — first-use graph capture, a code object loading, a
        profiler attaching).
        self_test: run self_test() at the end of set-up (collective).  Groups made by local_ranks
        are not self-tested: their ranks are driven by one caller, who checks them."""
        self.max_bytes = int(max_bytes)
        self.algorithm = int(algorithm)
        self.wait_seconds = float(wait_seconds) if wait_seconds is not None else self.DEFAULT_WAIT_SECONDS
        self.self_test_reduces = 0  # reduces the self-test has put on this communicator's count
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
                _capi.comm_set_wait_budget(self._comm, self.wait_seconds)
            return
        # Set-up succeeds or fails on ALL ranks: a rank that raised alone would leave the others
        # waiting in the next collective of the group.
        error: Optional[str] = None
        handles = b""
        try:
            with torch.cuda.device(self.device):
                self._comm, handles = _capi.comm_create(self.rank, self.world, self.max_bytes)
                _capi.comm_set_wait_budget(self._comm, self.wait_seconds)
            if self.world > 1 and not any(handles):
                error = ("hipIpcGetMemHandle exported no handle for the inbox" + self._ipc_hint())
        except Exception as e:  # noqa: BLE001
            error = f"{type(e).__name__}: {e}"
        gathered: List[Optional[bytes]] = [None] * self.world
        dist.all_gather_object(gathered, None if error else handles, group=group)
        if error is None and all(h is not None for h in gathered):
            try:
                with torch.cuda.device(self.device):
                    _capi.comm_connect(self._comm, gathered)
            except Exception as e:  # noqa: BLE001
                error = f"{type(e).__name__}: {e}" + self._ipc_hint()
        elif error is None:
            error = "another rank could not create its communicator"
        self._agree(error, "set-up")  # also the barrier: no rank reduces before every rank has mapped every inbox
        if self_test and self.world > 1:
            self.self_test()

    @staticmethod
    def _ipc_hint() -> str:
        import os
        if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0":
            return ""
        return (" (HSA_ENABLE_IPC_MODE_LEGACY is not 0 in this process: hosts whose driver only supports dmabuf IPC "
                "need HSA_ENABLE_IPC_MODE_LEGACY=0 in every rank's environment before the HIP runtime starts)")

    def _agree(self, error: Optional[str], what: str) -> None:
        """Collective: raises on every rank if any rank reports an error."""
        errors: List[Optional[str]] = [None] * self.world
        dist.all_gather_object(errors, error, group=self._group)
        failed = [(r, e) for r, e in enumerate(errors) if e]
        if failed:
            self.close()
            raise RuntimeError(f"one-shot all-reduce {what} failed on rank {failed[0][0]}: {failed[0][1]}")

    def _self_test_cases(self):
        """(elements, algorithm) of the reduces self_test runs: a small and a multi-MiB message (as far
        as max_bytes allows), one-shot and two-shot, fp16."""
        small = min(4096, self.max_bytes // 2 // 8 * 8)
        big = min(4 << 20, self.max_bytes) // 2 // 8 * 8
        sizes = [small] if big <= small else [small, big]
        return [(n, alg) for n in sizes if n > 0 for alg in (self.ONE_SHOT, self.TWO_SHOT)]

    def _self_test_once(self, n: int, algorithm: int, round_: int) -> Optional[str]:
        # every element differs by rank AND by position, so a vector that arrived late, from the
        # wrong slot or from the previous reduce cannot add up to the expected sum
        idx = torch.arange(n, device=self.device, dtype=torch.float32) % 61.0
        mine = (idx + float(self.rank + 1 + round_)).to(torch.float16)
        x = mine.clone()
        _capi.comm_all_reduce(self._comm, x, torch.cuda.current_stream(self.device), algorithm)
        torch.cuda.synchronize(self.device)
        want = torch.zeros(n, device=self.device, dtype=torch.float32)
        for r in range(self.world):  # rank order, fp32, then one rounding: what the kernel does
            want += (idx + float(r + 1 + round_)).to(torch.float16).float()
        want = want.to(torch.float16)
        gave_up = self.status()[1]
        if gave_up:
            return f"{gave_up} wait(s) given up (a peer did not arrive within {self.wait_seconds} s)"
        bad = int((x != want).sum())
        if bad:
            first = int((x != want).nonzero()[0])
            return (f"{bad} of {n} elements wrong (algorithm {algorithm}), first at {first}: got {float(x[first])}, "
                    f"expected {float(want[first])}")
        return None

    def self_test(self) -> None:
        """Collective: reduces of known values on every rank's current stream — both algorithms, a small
        and a multi-MiB message, twice each so that both inbox parities are used — checked element by
        element on every rank (exact sum, no wait given up); raises on every rank if any rank saw a
        wrong result.  Bounded by the wait budget even when peers cannot be reached."""
        error: Optional[str] = None
        try:
            with torch.cuda.device(self.device):
                for n, alg in self._self_test_cases():
                    for round_ in range(2):
                        if error is None:
                            error = self._self_test_once(n, alg, round_)
                            self.self_test_reduces += 1
        except Exception as e:  # noqa: BLE001
            error = f"{type(e).__name__}: {e}"
        self._agree(error, "self-test")

    @classmethod
    def local_ranks(cls, world: int, max_bytes: int, devices: Optional[Sequence[torch.device]] = None,
                    algorithm: int = _capi.COMM_AUTO, wait_seconds: Optional[float] = None) -> List["PushAllReduce"]:
        """All ranks of a group inside this process, connected by pointer.  devices: one per rank
        (default: the current device for every rank)."""
        devices = list(devices) if devices is not None else [torch.device("cuda", torch.cuda.current_device())] * world
        ranks = [cls(max_bytes, device=devices[r], algorithm=algorithm, wait_seconds=wait_seconds, _local=(r, world))
                 for r in range(world)]
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
        given up (a peer did not arrive within wait_seconds) left NaNs in this rank's result, and the
        communicator is broken from then on: replace it."""
        with torch.cuda.device(self.device):
            return _capi.comm_status(self._comm)

    def check(self) -> None:
        """Raises if any reduce of this rank gave up waiting for a peer (its result, and possibly
        every later one, is NaN).  Synchronises with the device: call it where the step does anyway."""
        done, gave_up = self.status()
        if gave_up:
            raise RuntimeError(f"one-shot all-reduce: rank {self.rank} gave up {gave_up} wait(s) for its peers "
                               f"(budget {self.wait_seconds} s) after {done} reduces; its results since then are NaN. "
                               "Rebuild the communicator, or fall back to the group's all_reduce.")

    def close(self) -> None:
        if self._comm is not None:
            with torch.cuda.device(self.device):
                torch.cuda.synchronize(self.device)
                _capi.comm_destroy(self._comm)
            self._comm = None
