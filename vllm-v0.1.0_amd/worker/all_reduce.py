"""Tensor-parallel all-reduce launcher for the step after the attention path.

Counterpart of GraphAllReduce and initialize_all_reduce_launcher
(vllm/model_executor/parallel_utils/parallel_state.py:191-203, 506-556), used by
RowParallelLinear after o_proj and down_proj
(parallel_utils/tensor_parallel/layers.py:416-426): a persistent
[max_num_tokens, hidden_size] buffer that the row-parallel matmul writes its
partial sums into, and `launch(x)` that sums a leading slice of it over the
tensor-parallel group in place.  `buffer` and `launch` keep that contract.

What is different, for one process per MI355X with RCCL (torch.distributed
backend "nccl") over xGMI:

  * The collective runs on a SIDE HIP stream that the launcher owns.  Hand-off
    is by events: the side stream waits for an event recorded on the caller's
    stream after the partial sums were written; the caller's stream waits for an
    event recorded on the side stream after the collective.  `launch_async`
    returns after the first half, `Pending.wait()` does the second, so work the
    caller queues in between (the next layer's cache write and attention in a
    decode step, or the matmul of the next token chunk) overlaps the collective.
    `launch` is the two halves back to back: the reference's semantics.
  * Two buffers in a ring (`next_buffer`): partial sums of reduce n + 1 can be
    written while reduce n is in flight.  A buffer is handed out again only
    after the caller's stream has been made to wait for its previous reduce;
    un-waited reduces are fenced automatically, so a caller cannot overwrite
    data that is still being summed.
  * Launch cost matters more than bandwidth (a decode step reduces
    256 x 4096 fp16 = 2 MiB twice per layer): each (buffer, token count) is
    replayed from a HIP graph captured ON THE SIDE STREAM, on first use or ahead
    of time by `precapture` (the reference captures max_num_tokens / 8 graphs in
    the constructor, several seconds of start-up per worker).

Graph capture needs a GPU; on CPU tensors (gloo groups, the tests) and with
disable_graph=True the collective is issued directly, and the same hand-off
protocol is tracked by sequence numbers so that its ordering rules are checked
without a device.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


class Pending:
    """A reduce that was handed to the side stream.  wait() makes the caller's
    current stream wait for it and returns the summed slice."""

    def __init__(self, launcher: "GraphAllReduce", index: int, x: torch.Tensor, seq: int) -> None:
        self._launcher = launcher
        self._index = index
        self._x = x
        self._seq = seq
        self.waited = False

    def wait(self) -> torch.Tensor:
        if not self.waited:
            self._launcher._wait(self._index, self._seq)
            self.waited = True
        return self._x


class GraphAllReduce:

    NUM_BUFFERS = 2

    def __init__(self, max_num_tokens: int, hidden_size: int, dtype: torch.dtype, disable_graph: bool = False,
                 group: Optional[dist.ProcessGroup] = None, device: Optional[torch.device] = None,
                 reduce_fn: Optional[Callable[[torch.Tensor], None]] = None, record_trace: bool = False) -> None:
        """reduce_fn(t): the in-place collective; default = all_reduce(SUM) over `group`.  (The
        tests pass a local stand-in to drive the stream and graph machinery on one GPU.)
        record_trace: keep the hand-off protocol as it ran in `self.trace` (tests only: the list
        grows with every launch)."""
        self.max_num_tokens = max_num_tokens
        self.hidden_size = hidden_size
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        if device is None:
            # the group's backend decides where its tensors live: RCCL ("nccl") sums device memory, gloo host memory
            on_gpu = torch.cuda.is_available() and (not dist.is_initialized() or dist.get_backend(group) == "nccl")
            device = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.disable_graph = disable_graph or not self.on_gpu
        self._reduce_fn = reduce_fn if reduce_fn is not None else (lambda t: dist.all_reduce(t, group=self.group))
        self.graphs: Dict[Tuple[int, int], "torch.cuda.CUDAGraph"] = {}
        # (what, buffer index, sequence number) of every hand-off step, if asked for
        self.trace: Optional[List[Tuple[str, int, int]]] = [] if record_trace else None
        self._active = self.world_size > 1 or reduce_fn is not None
        if not self._active:
            # a single rank never reduces (layers.py:411-413 skips the launcher)
            self.buffer = None
            self.buffers = []
            return
        self.buffers = [torch.empty(max_num_tokens, hidden_size, dtype=dtype, device=self.device)
                        for _ in range(self.NUM_BUFFERS)]
        self.buffer = self.buffers[0]  # the reference's attribute: what RowParallelLinear slices
        self._cursor = 0
        self._seq = 0
        self._in_flight: List[Optional[Pending]] = [None] * self.NUM_BUFFERS
        if self.on_gpu:
            self.comm_stream = torch.cuda.Stream(device=self.device)
            self._ready = [torch.cuda.Event() for _ in range(self.NUM_BUFFERS)]
            self._done = [torch.cuda.Event() for _ in range(self.NUM_BUFFERS)]

    def _log(self, what: str, index: int, seq: int) -> None:
        if self.trace is not None:
            self.trace.append((what, index, seq))

    # ---- buffers ------------------------------------------------------------------

    def _index_of(self, x: torch.Tensor) -> int:
        for i, b in enumerate(self.buffers):
            if x.data_ptr() == b.data_ptr() and x.shape[1:] == b.shape[1:] and x.shape[0] <= self.max_num_tokens:
                return i
        raise ValueError("launch() takes a leading slice of one of the launcher's buffers")

    def next_buffer(self, num_tokens: int) -> torch.Tensor:
        """The next buffer of the ring, [num_tokens, hidden]: safe to write on the current stream
        (its previous reduce, if still un-waited, is waited for here)."""
        if num_tokens > self.max_num_tokens:
            raise ValueError(f"{num_tokens} tokens exceed the launcher's {self.max_num_tokens}")
        i = self._cursor
        self._cursor = (i + 1) % self.NUM_BUFFERS
        pending = self._in_flight[i]
        if pending is not None:
            pending.wait()
        return self.buffers[i][:num_tokens]

    # ---- graphs -------------------------------------------------------------------

    def _build_graph(self, index: int, num_tokens: int) -> "torch.cuda.CUDAGraph":
        # Runs with the side stream current.  Warm-up outside the capture (communicator set-up
        # allocates) on live data: the caller's partial sums must survive until the replay.
        x = self.buffers[index][:num_tokens]
        keep = x.clone()
        self._reduce_fn(x)
        self.comm_stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=self.comm_stream):
            self._reduce_fn(x)
        x.copy_(keep)
        self.comm_stream.synchronize()
        return graph

    def precapture(self, token_counts: Iterable[int]) -> None:
        """Captures the graphs of these token counts for every buffer now, so that no capture
        (a device synchronisation and an allocation) happens inside the serving loop.  The
        reference's constructor does this for every multiple of 8 (parallel_state.py:531-534).
        Collective: every rank must call it with the same counts."""
        if not self._active or self.disable_graph:
            return
        torch.cuda.current_stream(self.device).synchronize()
        with torch.cuda.stream(self.comm_stream):
            for n in token_counts:
                for i in range(self.NUM_BUFFERS):
                    if (i, n) not in self.graphs and 0 < n <= self.max_num_tokens:
                        self.graphs[(i, n)] = self._build_graph(i, n)
        self.comm_stream.synchronize()

    # ---- launch -------------------------------------------------------------------

    def launch_async(self, x: torch.Tensor) -> Pending:
        """x: a leading slice of one of the launcher's buffers, already written on the current
        stream.  Queues its in-place sum over the group on the side stream and returns; the
        current stream is not blocked until Pending.wait()."""
        if not self._active:
            p = Pending(self, -1, x, -1)
            p.waited = True
            return p
        i = self._index_of(x)
        previous = self._in_flight[i]
        if previous is not None and not previous.waited:
            raise RuntimeError("this buffer's previous reduce was never waited for: its partial sums were "
                               "overwritten while still being summed (use next_buffer(), or wait() first)")
        num_tokens = x.shape[0]
        self._seq += 1
        seq = self._seq
        self._log("ready", i, seq)
        if self.on_gpu:
            self._ready[i].record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(self._ready[i])
                if self.disable_graph:
                    self._reduce_fn(x)
                else:
                    graph = self.graphs.get((i, num_tokens))
                    if graph is None:
                        graph = self.graphs[(i, num_tokens)] = self._build_graph(i, num_tokens)
                    graph.replay()
                self._done[i].record(self.comm_stream)
        else:
            self._reduce_fn(x)
        self._log("reduced", i, seq)
        pending = Pending(self, i, x, seq)
        self._in_flight[i] = pending
        return pending

    def _wait(self, index: int, seq: int) -> None:
        if self.on_gpu:
            torch.cuda.current_stream(self.device).wait_event(self._done[index])
        self._log("waited", index, seq)
        if self._in_flight[index] is not None and self._in_flight[index]._seq == seq:
            self._in_flight[index] = None

    def check(self) -> None:
        """Health of the collective behind the launcher: a reduce_fn that can fail without raising
        (PushAllReduce: a peer that never arrived leaves NaNs) offers check(), which raises.  It
        synchronises with the device, so the step loop calls it where it synchronises anyway (after
        the step's tokens were sampled), before the step's results are used."""
        fn_check = getattr(self._reduce_fn, "check", None)
        if self._active and callable(fn_check):
            fn_check()

    def launch(self, x: torch.Tensor) -> torch.Tensor:
        """x must be a leading slice of `self.buffer` (or of a ring buffer); summed over the
        group in place, ordered on the current stream like any other operation on it."""
        if not self._active:
            return x
        return self.launch_async(x).wait()


_ALL_REDUCE_LAUNCHER: Optional[GraphAllReduce] = None


def initialize_all_reduce_launcher(max_num_tokens: int, hidden_size: int, dtype: torch.dtype,
                                   disable_graph: bool = False, group: Optional[dist.ProcessGroup] = None,
                                   device: Optional[torch.device] = None) -> None:
    global _ALL_REDUCE_LAUNCHER
    _ALL_REDUCE_LAUNCHER = GraphAllReduce(max_num_tokens, hidden_size, dtype, disable_graph, group, device)


def get_all_reduce_launcher() -> GraphAllReduce:
    assert _ALL_REDUCE_LAUNCHER is not None, "all reduce launcher is not initialized"
    return _ALL_REDUCE_LAUNCHER
