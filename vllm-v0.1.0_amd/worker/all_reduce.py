"""Tensor-parallel all-reduce launcher for the step after the attention path.

Counterpart of GraphAllReduce and initialize_all_reduce_launcher
(vllm/model_executor/parallel_utils/parallel_state.py:191-203, 506-556), used by
RowParallelLinear after o_proj and down_proj
(parallel_utils/tensor_parallel/layers.py:416-426): one persistent
[max_num_tokens, hidden_size] buffer that the matmul writes into, and
`launch(x)` that sums a leading slice of it over the tensor-parallel group in
place.

One process per GPU; the collective is RCCL (torch.distributed backend "nccl")
over xGMI.  Launch cost matters more than bandwidth here: a decode step reduces
256 x 4096 fp16 = 2 MiB twice per layer.  The launch is therefore replayed from
a HIP graph per token count, captured on first use (the reference captures
max_num_tokens / 8 graphs up front, several seconds of start-up per worker).
Graph capture needs a GPU process group; on CPU groups (gloo, the tests) and
with disable_graph=True the collective is issued directly.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist


class GraphAllReduce:

    def __init__(self, max_num_tokens: int, hidden_size: int, dtype: torch.dtype, disable_graph: bool = False,
                 group: Optional[dist.ProcessGroup] = None, device: Optional[torch.device] = None) -> None:
        self.max_num_tokens = max_num_tokens
        self.hidden_size = hidden_size
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self.disable_graph = disable_graph or self.device.type != "cuda"
        self.graphs: Dict[int, "torch.cuda.CUDAGraph"] = {}
        if self.world_size == 1:
            # a single rank never reduces (layers.py:411-413 skips the launcher)
            self.buffer = None
            return
        self.buffer = torch.empty(max_num_tokens, hidden_size, dtype=dtype, device=self.device)

    def _build_graph(self, num_tokens: int) -> "torch.cuda.CUDAGraph":
        # Warm-up outside the capture (communicator set-up allocates) on a copy:
        # the caller's partial sums in the buffer must survive until the replay.
        keep = self.buffer[:num_tokens].clone()
        dist.all_reduce(self.buffer[:num_tokens], group=self.group)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            dist.all_reduce(self.buffer[:num_tokens], group=self.group)
        self.buffer[:num_tokens].copy_(keep)
        torch.cuda.synchronize(self.device)
        return graph

    def launch(self, x: torch.Tensor) -> torch.Tensor:
        """x must be `self.buffer[:num_tokens]`; summed over the group in place."""
        if self.world_size == 1:
            return x
        num_tokens = x.shape[0]
        if x.data_ptr() != self.buffer.data_ptr() or x.shape[1:] != self.buffer.shape[1:] or num_tokens > self.max_num_tokens:
            raise ValueError("launch() takes a leading slice of the launcher's buffer")
        if self.disable_graph:
            dist.all_reduce(x, group=self.group)
            return x
        graph = self.graphs.get(num_tokens)
        if graph is None:
            graph = self.graphs[num_tokens] = self._build_graph(num_tokens)
        graph.replay()
        return x


_ALL_REDUCE_LAUNCHER: Optional[GraphAllReduce] = None


def initialize_all_reduce_launcher(max_num_tokens: int, hidden_size: int, dtype: torch.dtype,
                                   disable_graph: bool = False, group: Optional[dist.ProcessGroup] = None,
                                   device: Optional[torch.device] = None) -> None:
    global _ALL_REDUCE_LAUNCHER
    _ALL_REDUCE_LAUNCHER = GraphAllReduce(max_num_tokens, hidden_size, dtype, disable_graph, group, device)


def get_all_reduce_launcher() -> GraphAllReduce:
    assert _ALL_REDUCE_LAUNCHER is not None, "all reduce launcher is not initialized"
    return _ALL_REDUCE_LAUNCHER
