"""ctypes binding of libpaged_attn_mi355x.so (the C ABI in
include/paged_attn_mi355x.h).

This is the raw boundary: pointers and sizes only.  The torch-facing operator
modules (`attention_ops`, `cache_ops`, `pos_encoding_ops`, `layernorm_ops`, `activation_ops`) are separate pybind11 extensions on top
of the same library; this binding exists so that the C ABI itself can be
driven and tested from Python exactly as a foreign-language host would drive
it.  There is no fallback: if the library is missing, importing fails.
"""
from __future__ import annotations

import ctypes
import os
from typing import Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpaged_attn_mi355x.so")

PA_OK = 0
PA_ERR_INVALID_ARGUMENT = 1
PA_ERR_UNSUPPORTED = 2
PA_ERR_HIP = 3
PA_DTYPE_F32, PA_DTYPE_F16, PA_DTYPE_BF16 = 0, 1, 2
PA_COPY_DEVICE_TO_DEVICE, PA_COPY_DEVICE_TO_HOST, PA_COPY_HOST_TO_DEVICE = 0, 1, 2
PA_ABI_VERSION = 1

EXPORTS = (
    "pa_abi_version",
    "pa_last_error",
    "pa_single_query_cached_kv_attention",
    "pa_single_query_cached_kv_attention_ws",
    "pa_single_query_cached_kv_attention_planned",
    "pa_single_query_cached_kv_attention_static",
    "pa_append_single_query_cached_kv_attention",
    "pa_decode_workspace_bytes",
    "pa_multi_query_kv_attention",
    "pa_reshape_and_cache",
    "pa_rotary_embedding_neox_reshape_and_cache",
    "pa_gather_cached_kv",
    "pa_copy_blocks",
    "pa_swap_blocks",
    "pa_rotary_embedding_neox",
    "pa_rms_norm",
    "pa_silu_and_mul",
    "pa_comm_create",
    "pa_comm_connect",
    "pa_comm_connect_local",
    "pa_comm_all_reduce",
    "pa_comm_set_wait_budget",
    "pa_comm_status",
    "pa_comm_destroy",
)


class PagedAttnError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[paged_attn_mi355x code {code}] {message}")
        self.code = code


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C vllm-v0.1.0_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback for this path.")
    # torch carries its own HIP runtime; load torch first so that this library
    # binds to the runtime the process already uses instead of a second copy.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    lib.pa_abi_version.argtypes = []
    lib.pa_abi_version.restype = i32
    lib.pa_last_error.argtypes = []
    lib.pa_last_error.restype = ctypes.c_char_p
    lib.pa_single_query_cached_kv_attention.argtypes = [
        vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, vp]
    lib.pa_single_query_cached_kv_attention.restype = i32
    lib.pa_single_query_cached_kv_attention_static.argtypes = lib.pa_single_query_cached_kv_attention.argtypes
    lib.pa_single_query_cached_kv_attention_static.restype = i32
    lib.pa_single_query_cached_kv_attention_ws.argtypes = [
        vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, vp, ctypes.c_size_t, vp]
    lib.pa_single_query_cached_kv_attention_ws.restype = i32
    lib.pa_single_query_cached_kv_attention_planned.argtypes = [
        vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, vp, ctypes.c_size_t, i32, i64, vp]
    lib.pa_single_query_cached_kv_attention_planned.restype = i32
    lib.pa_append_single_query_cached_kv_attention.argtypes = [
        vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, i64, i64, i64, i32, vp, ctypes.c_size_t, i32,
        i64, vp]
    lib.pa_append_single_query_cached_kv_attention.restype = i32
    lib.pa_decode_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    lib.pa_decode_workspace_bytes.restype = ctypes.c_size_t
    lib.pa_multi_query_kv_attention.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f32, i64, i64, i64, i32, vp]
    lib.pa_multi_query_kv_attention.restype = i32
    for fn in (lib.pa_reshape_and_cache, lib.pa_gather_cached_kv):
        fn.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, vp]
        fn.restype = i32
    lib.pa_rotary_embedding_neox_reshape_and_cache.argtypes = [
        vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i64, i64, i32, vp]
    lib.pa_rotary_embedding_neox_reshape_and_cache.restype = i32
    lib.pa_copy_blocks.argtypes = [vp, vp, i32, vp, i32, i64, vp]
    lib.pa_copy_blocks.restype = i32
    lib.pa_swap_blocks.argtypes = [vp, vp, vp, i32, i64, i32, vp]
    lib.pa_swap_blocks.restype = i32
    lib.pa_rotary_embedding_neox.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, vp]
    lib.pa_rotary_embedding_neox.restype = i32
    lib.pa_rms_norm.argtypes = [vp, vp, vp, f32, i32, i32, i32, vp]
    lib.pa_rms_norm.restype = i32
    lib.pa_silu_and_mul.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.pa_silu_and_mul.restype = i32
    lib.pa_comm_create.argtypes = [i32, i32, ctypes.c_size_t, ctypes.POINTER(vp), vp]
    lib.pa_comm_create.restype = i32
    lib.pa_comm_connect.argtypes = [vp, vp]
    lib.pa_comm_connect.restype = i32
    lib.pa_comm_connect_local.argtypes = [vp, ctypes.POINTER(vp)]
    lib.pa_comm_connect_local.restype = i32
    lib.pa_comm_all_reduce.argtypes = [vp, vp, ctypes.c_size_t, i32, i32, vp]
    lib.pa_comm_all_reduce.restype = i32
    lib.pa_comm_set_wait_budget.argtypes = [vp, ctypes.c_double]
    lib.pa_comm_set_wait_budget.restype = i32
    lib.pa_comm_status.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    lib.pa_comm_status.restype = i32
    lib.pa_comm_destroy.argtypes = [vp]
    lib.pa_comm_destroy.restype = i32
    if lib.pa_abi_version() != PA_ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.pa_abi_version()} != {PA_ABI_VERSION}")
    return lib


lib = _load()


def last_error() -> str:
    return lib.pa_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != PA_OK:
        raise PagedAttnError(rc, last_error())


def dtype_code(torch_dtype) -> int:
    import torch
    try:
        return {torch.float32: PA_DTYPE_F32, torch.float16: PA_DTYPE_F16, torch.bfloat16: PA_DTYPE_BF16}[torch_dtype]
    except KeyError:
        return -1


def _stream_handle(stream) -> ctypes.c_void_p:
    if stream is None:
        import torch
        stream = torch.cuda.current_stream()
    return ctypes.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))


# ---- thin helpers over torch tensors: every argument is reduced to a raw
# pointer or integer before the call, nothing else is done here -------------

def single_query_cached_kv_attention(out, query, key_cache, value_cache, scale, block_tables,
                                     context_lens, block_size, max_context_len, stream=None,
                                     dtype=None, head_size=None) -> None:
    S, H, D = query.shape
    rc = lib.pa_single_query_cached_kv_attention(
        out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), float(scale),
        block_tables.data_ptr(), context_lens.data_ptr(), S, H, D if head_size is None else head_size,
        block_size, block_tables.shape[1], max_context_len, query.stride(0) if S > 0 else 0,
        dtype_code(query.dtype) if dtype is None else dtype, _stream_handle(stream))
    check(rc)


DECODE_STATIC_SCRATCH_BYTES = 32 << 20  # PA_DECODE_STATIC_SCRATCH_BYTES


def single_query_cached_kv_attention_static(out, query, key_cache, value_cache, scale, block_tables,
                                            context_lens, block_size, max_context_len, stream=None) -> None:
    """The _ws form on the library's own per-GPU scratch array: allocates nothing."""
    S, H, D = query.shape
    rc = lib.pa_single_query_cached_kv_attention_static(
        out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), float(scale),
        block_tables.data_ptr(), context_lens.data_ptr(), S, H, D, block_size, block_tables.shape[1],
        max_context_len, query.stride(0) if S > 0 else 0, dtype_code(query.dtype), _stream_handle(stream))
    check(rc)


def decode_workspace_bytes(num_seqs, num_heads, head_size, block_size, max_context_len) -> int:
    return int(lib.pa_decode_workspace_bytes(num_seqs, num_heads, head_size, block_size, max_context_len))


def single_query_cached_kv_attention_ws(out, query, key_cache, value_cache, scale, block_tables,
                                        context_lens, block_size, max_context_len, workspace,
                                        stream=None) -> None:
    """`workspace` is a device tensor (any dtype) or None."""
    S, H, D = query.shape
    ws_ptr = workspace.data_ptr() if workspace is not None else None
    ws_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    rc = lib.pa_single_query_cached_kv_attention_ws(
        out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), float(scale),
        block_tables.data_ptr(), context_lens.data_ptr(), S, H, D, block_size, block_tables.shape[1],
        max_context_len, query.stride(0) if S > 0 else 0, dtype_code(query.dtype), ws_ptr, ws_bytes,
        _stream_handle(stream))
    check(rc)


def single_query_cached_kv_attention_planned(out, query, key_cache, value_cache, scale, block_tables,
                                             context_lens, block_size, max_context_len, workspace,
                                             plan_ready, total_context_len=-1, stream=None) -> None:
    """One workspace for all calls on the same context_lens: plan_ready False on the first, True after."""
    S, H, D = query.shape
    ws_ptr = workspace.data_ptr() if workspace is not None else None
    ws_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    rc = lib.pa_single_query_cached_kv_attention_planned(
        out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), float(scale),
        block_tables.data_ptr(), context_lens.data_ptr(), S, H, D, block_size, block_tables.shape[1],
        max_context_len, query.stride(0) if S > 0 else 0, dtype_code(query.dtype), ws_ptr, ws_bytes,
        1 if plan_ready else 0, int(total_context_len), _stream_handle(stream))
    check(rc)


def append_single_query_cached_kv_attention(out, query, key, value, key_cache, value_cache, scale, block_tables,
                                            context_lens, block_size, max_context_len, workspace=None, plan_ready=0,
                                            total_context_len=-1, stream=None) -> None:
    """reshape_and_cache of each sequence's new token (key / value [S, H, D]) at the last position of its
    context, and the decode attention over that context, in one launch.  plan_ready: 0 make a shared plan,
    1 plan is ready, -1 plan for this call alone."""
    S, H, D = query.shape
    ws_ptr = workspace.data_ptr() if workspace is not None else None
    ws_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    rc = lib.pa_append_single_query_cached_kv_attention(
        out.data_ptr(), query.data_ptr(), key.data_ptr(), value.data_ptr(), key_cache.data_ptr(),
        value_cache.data_ptr(), float(scale), block_tables.data_ptr(), context_lens.data_ptr(), S, H, D, block_size,
        block_tables.shape[1], max_context_len, query.stride(0) if S > 0 else 0, key.stride(0) if S > 0 else 0,
        value.stride(0) if S > 0 else 0, dtype_code(query.dtype), ws_ptr, ws_bytes, int(plan_ready),
        int(total_context_len), _stream_handle(stream))
    check(rc)


def multi_query_kv_attention(out, query, key, value, cu_seqlens: Sequence[int], scale, stream=None) -> None:
    """query / key / value [T, H, D] (row strides free), out dense; cu_seqlens is a host sequence."""
    T, H, D = query.shape
    cu = (ctypes.c_int32 * len(cu_seqlens))(*[int(x) for x in cu_seqlens])
    rc = lib.pa_multi_query_kv_attention(
        out.data_ptr(), query.data_ptr(), key.data_ptr(), value.data_ptr(), cu, len(cu_seqlens) - 1, H, D,
        float(scale), query.stride(0) if T > 0 else H * D, key.stride(0) if T > 0 else H * D,
        value.stride(0) if T > 0 else H * D, dtype_code(query.dtype), _stream_handle(stream))
    check(rc)


def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, stream=None) -> None:
    T, H, D = key.shape
    rc = lib.pa_reshape_and_cache(
        key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
        slot_mapping.data_ptr(), T, H, D, value_cache.shape[3], key.stride(0) if T > 0 else 0,
        value.stride(0) if T > 0 else 0, dtype_code(key.dtype), _stream_handle(stream))
    check(rc)


def rotary_embedding_neox_reshape_and_cache(positions, query, key, value, key_cache, value_cache, slot_mapping,
                                            cos_sin_cache, stream=None) -> None:
    """query / key / value: [T, H, D] views (row strides free); rotates query and key in place and
    writes the rotated key and the value into the caches."""
    T, H, D = key.shape
    rc = lib.pa_rotary_embedding_neox_reshape_and_cache(
        positions.data_ptr(), query.data_ptr(), key.data_ptr(), value.data_ptr(), cos_sin_cache.data_ptr(),
        key_cache.data_ptr(), value_cache.data_ptr(), slot_mapping.data_ptr(), T, H, D, cos_sin_cache.shape[1],
        value_cache.shape[3], query.stride(0) if T > 0 else 0, key.stride(0) if T > 0 else 0,
        value.stride(0) if T > 0 else 0, dtype_code(key.dtype), _stream_handle(stream))
    check(rc)


def gather_cached_kv(key, value, key_cache, value_cache, slot_mapping, stream=None) -> None:
    T, H, D = key.shape
    rc = lib.pa_gather_cached_kv(
        key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
        slot_mapping.data_ptr(), T, H, D, value_cache.shape[3], key.stride(0) if T > 0 else 0,
        value.stride(0) if T > 0 else 0, dtype_code(key.dtype), _stream_handle(stream))
    check(rc)


def _pairs(src: Sequence[int], dst: Sequence[int]):
    n = len(src)
    arr = (ctypes.c_int64 * (2 * n))()
    for i in range(n):
        arr[2 * i] = int(src[i])
        arr[2 * i + 1] = int(dst[i])
    return arr


def copy_blocks(key_caches, value_caches, src: Sequence[int], dst: Sequence[int], stream=None) -> None:
    L = len(key_caches)
    kp = (ctypes.c_void_p * max(L, 1))(*[t.data_ptr() for t in key_caches])
    vp = (ctypes.c_void_p * max(L, 1))(*[t.data_ptr() for t in value_caches])
    block_bytes = key_caches[0][0].numel() * key_caches[0].element_size() if L else 0
    rc = lib.pa_copy_blocks(kp, vp, L, _pairs(src, dst), len(src), block_bytes, _stream_handle(stream))
    check(rc)


def swap_blocks(src_cache, dst_cache, src: Sequence[int], dst: Sequence[int], stream=None) -> None:
    if src_cache.is_cuda and dst_cache.is_cuda:
        kind = PA_COPY_DEVICE_TO_DEVICE
    elif src_cache.is_cuda:
        kind = PA_COPY_DEVICE_TO_HOST
    else:
        kind = PA_COPY_HOST_TO_DEVICE
    block_bytes = src_cache[0].numel() * src_cache.element_size()
    rc = lib.pa_swap_blocks(src_cache.data_ptr(), dst_cache.data_ptr(), _pairs(src, dst), len(src),
                            block_bytes, kind, _stream_handle(stream))
    check(rc)


def rotary_embedding_neox(positions, query, key, head_size, cos_sin_cache, stream=None) -> None:
    T = query.shape[0]
    rc = lib.pa_rotary_embedding_neox(
        positions.data_ptr(), query.data_ptr(), key.data_ptr(), cos_sin_cache.data_ptr(), T,
        query.shape[1] // head_size, head_size, cos_sin_cache.shape[1], query.stride(0) if T > 0 else 0,
        key.stride(0) if T > 0 else 0, dtype_code(query.dtype), _stream_handle(stream))
    check(rc)


def rms_norm(out, input, weight, epsilon, stream=None) -> None:
    rc = lib.pa_rms_norm(out.data_ptr(), input.data_ptr(), weight.data_ptr(), float(epsilon), input.shape[0],
                         input.shape[1], dtype_code(input.dtype), _stream_handle(stream))
    check(rc)


def silu_and_mul(out, input, stream=None) -> None:
    rc = lib.pa_silu_and_mul(out.data_ptr(), input.data_ptr(), input.shape[0], input.shape[1] // 2,
                             dtype_code(input.dtype), _stream_handle(stream))
    check(rc)


# ---- one-shot all-reduce communicator ---------------------------------------------

COMM_HANDLE_BYTES = 128


def comm_create(rank: int, world: int, max_bytes: int):
    """On the current device.  Returns (communicator handle, bytes of IPC handles for peers in other processes)."""
    comm = ctypes.c_void_p()
    blob = (ctypes.c_ubyte * COMM_HANDLE_BYTES)()
    check(lib.pa_comm_create(int(rank), int(world), int(max_bytes), ctypes.byref(comm), blob))
    return comm, bytes(blob)


def comm_connect(comm, all_handles: Sequence[bytes]) -> None:
    """all_handles: every rank's handle bytes, in rank order."""
    blob = b"".join(all_handles)
    buf = (ctypes.c_ubyte * len(blob)).from_buffer_copy(blob)
    check(lib.pa_comm_connect(comm, buf))


def comm_connect_local(comm, all_comms) -> None:
    """all_comms: the communicators of all ranks (they live in this process), in rank order."""
    arr = (ctypes.c_void_p * len(all_comms))(*[c.value for c in all_comms])
    check(lib.pa_comm_connect_local(comm, arr))


COMM_AUTO, COMM_ONE_SHOT, COMM_TWO_SHOT = 0, 1, 2


def comm_all_reduce(comm, tensor, stream=None, algorithm: int = COMM_AUTO) -> None:
    """In-place sum of a contiguous tensor over the communicator's ranks, on `stream` (default: torch's current)."""
    check(lib.pa_comm_all_reduce(comm, tensor.data_ptr(), tensor.numel(), dtype_code(tensor.dtype), int(algorithm),
                                 _stream_handle(stream)))


def comm_set_wait_budget(comm, seconds: float) -> None:
    """How long a reduce launched (or captured) from now on waits for its peers before it gives up."""
    check(lib.pa_comm_set_wait_budget(comm, float(seconds)))


def comm_status(comm):
    """(reduces completed, timeouts); synchronises with the device."""
    done, timeouts = ctypes.c_uint32(), ctypes.c_uint32()
    check(lib.pa_comm_status(comm, ctypes.byref(done), ctypes.byref(timeouts)))
    return done.value, timeouts.value


def comm_destroy(comm) -> None:
    check(lib.pa_comm_destroy(comm))
