"""ctypes front end of the C oracle (oracle/pa_oracle.c).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module.  It operates on numpy
arrays in storage form (float32 / float16 / uint16 bfloat16 bit patterns).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpa_oracle.so")
_lib = None

DTYPE_CODE = {"float32": 0, "float16": 1, "bfloat16": 2}


def build(force: bool = False) -> str:
    """Compiles libpa_oracle.so with gcc if missing or stale."""
    src = os.path.join(_HERE, "pa_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpa_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
        L.pao_single_query_cached_kv_attention.argtypes = [
            vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i64, i32, i32]
        L.pao_single_query_cached_kv_attention.restype = i32
        for fn in (L.pao_reshape_and_cache, L.pao_gather_cached_kv):
            fn.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32]
            fn.restype = i32
        L.pao_copy_blocks.argtypes = [vp, vp, i32, vp, i32, i64]
        L.pao_copy_blocks.restype = i32
        L.pao_swap_blocks.argtypes = [vp, vp, vp, i32, i64]
        L.pao_swap_blocks.restype = i32
        L.pao_rotary_embedding_neox.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32]
        L.pao_rotary_embedding_neox.restype = i32
        L.pao_rms_norm.argtypes = [vp, vp, vp, f32, i32, i32, i32]
        L.pao_rms_norm.restype = i32
        L.pao_silu_and_mul.argtypes = [vp, vp, i32, i32, i32]
        L.pao_silu_and_mul.restype = i32
        L.pao_multi_query_kv_attention.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f32, i64, i64, i64, i32]
        L.pao_multi_query_kv_attention.restype = i32
        L.pao_test_half_to_float.argtypes = [ctypes.c_uint16]
        L.pao_test_half_to_float.restype = f32
        L.pao_test_float_to_half.argtypes = [f32]
        L.pao_test_float_to_half.restype = ctypes.c_uint16
        L.pao_test_bf16_to_float.argtypes = [ctypes.c_uint16]
        L.pao_test_bf16_to_float.restype = f32
        L.pao_test_float_to_bf16.argtypes = [f32]
        L.pao_test_float_to_bf16.restype = ctypes.c_uint16
        _lib = L
    return _lib


def _ptr(a: np.ndarray) -> ctypes.c_void_p:
    return ctypes.c_void_p(a.ctypes.data)


def _elem_stride(a: np.ndarray, axis: int) -> int:
    assert a.strides[axis] % a.itemsize == 0
    return a.strides[axis] // a.itemsize


def _check_inner_dense(a: np.ndarray) -> None:
    """[T, H, D] array whose last two dims are dense (row stride is free)."""
    assert a.ndim == 3
    assert a.strides[2] == a.itemsize and a.strides[1] == a.shape[2] * a.itemsize


def single_query_cached_kv_attention(query: np.ndarray, key_cache: np.ndarray,
                                     value_cache: np.ndarray, scale: float,
                                     block_tables: np.ndarray, context_lens: np.ndarray,
                                     dtype: str, mode: int = 0) -> np.ndarray:
    """Returns out [S, H, D] in storage form.  query may be a strided view
    ([S, H, D] with arbitrary row stride)."""
    S, H, D = query.shape
    _check_inner_dense(query)
    B = value_cache.shape[3]
    assert key_cache.flags.c_contiguous and value_cache.flags.c_contiguous
    block_tables = np.ascontiguousarray(block_tables, dtype=np.int32)
    context_lens = np.ascontiguousarray(context_lens, dtype=np.int32)
    out = np.zeros((S, H, D), dtype=query.dtype)
    if S == 0:
        return out
    rc = lib().pao_single_query_cached_kv_attention(
        _ptr(out), _ptr(query), _ptr(key_cache), _ptr(value_cache), float(scale),
        _ptr(block_tables), _ptr(context_lens), S, H, D, B, block_tables.shape[1],
        _elem_stride(query, 0), DTYPE_CODE[dtype], mode)
    if rc != 0:
        raise RuntimeError(f"pa_oracle decode failed with code {rc}")
    return out


def reshape_and_cache(key: np.ndarray, value: np.ndarray, key_cache: np.ndarray,
                      value_cache: np.ndarray, slot_mapping: np.ndarray, dtype: str) -> None:
    """In place on key_cache / value_cache."""
    T, H, D = key.shape
    _check_inner_dense(key)
    _check_inner_dense(value)
    assert key_cache.flags.c_contiguous and value_cache.flags.c_contiguous
    slot_mapping = np.ascontiguousarray(slot_mapping, dtype=np.int32)
    if T == 0:
        return
    rc = lib().pao_reshape_and_cache(
        _ptr(key), _ptr(value), _ptr(key_cache), _ptr(value_cache), _ptr(slot_mapping),
        T, H, D, value_cache.shape[3], _elem_stride(key, 0), _elem_stride(value, 0),
        DTYPE_CODE[dtype])
    if rc != 0:
        raise RuntimeError(f"pa_oracle reshape_and_cache failed with code {rc}")


def gather_cached_kv(key: np.ndarray, value: np.ndarray, key_cache: np.ndarray,
                     value_cache: np.ndarray, slot_mapping: np.ndarray, dtype: str) -> None:
    """In place on key / value (strided [T, H, D] views are fine)."""
    T, H, D = key.shape
    _check_inner_dense(key)
    _check_inner_dense(value)
    slot_mapping = np.ascontiguousarray(slot_mapping, dtype=np.int32)
    if T == 0:
        return
    rc = lib().pao_gather_cached_kv(
        _ptr(key), _ptr(value), _ptr(key_cache), _ptr(value_cache), _ptr(slot_mapping),
        T, H, D, value_cache.shape[3], _elem_stride(key, 0), _elem_stride(value, 0),
        DTYPE_CODE[dtype])
    if rc != 0:
        raise RuntimeError(f"pa_oracle gather_cached_kv failed with code {rc}")


def _pairs(src: Sequence[int], dst: Sequence[int]) -> np.ndarray:
    p = np.empty(2 * len(src), dtype=np.int64)
    p[0::2] = np.asarray(src, dtype=np.int64)
    p[1::2] = np.asarray(dst, dtype=np.int64)
    return p


def copy_blocks(key_caches: List[np.ndarray], value_caches: List[np.ndarray],
                src: Sequence[int], dst: Sequence[int]) -> None:
    L = len(key_caches)
    if L == 0 or len(src) == 0:
        return
    block_bytes = key_caches[0][0].nbytes
    kp = (ctypes.c_void_p * L)(*[a.ctypes.data for a in key_caches])
    vp = (ctypes.c_void_p * L)(*[a.ctypes.data for a in value_caches])
    pairs = _pairs(src, dst)
    rc = lib().pao_copy_blocks(kp, vp, L, _ptr(pairs), len(src), block_bytes)
    if rc != 0:
        raise RuntimeError(f"pa_oracle copy_blocks failed with code {rc}")


def swap_blocks(src_cache: np.ndarray, dst_cache: np.ndarray,
                src: Sequence[int], dst: Sequence[int]) -> None:
    if len(src) == 0:
        return
    pairs = _pairs(src, dst)
    rc = lib().pao_swap_blocks(_ptr(src_cache), _ptr(dst_cache), _ptr(pairs), len(src),
                               src_cache[0].nbytes)
    if rc != 0:
        raise RuntimeError(f"pa_oracle swap_blocks failed with code {rc}")


def rotary_embedding_neox(positions: np.ndarray, query: np.ndarray, key: np.ndarray, head_size: int,
                          cos_sin_cache: np.ndarray, dtype: str) -> None:
    """In place on query / key ([T, H * D] arrays, row stride free)."""
    T = query.shape[0]
    positions = np.ascontiguousarray(positions, dtype=np.int64)
    assert cos_sin_cache.flags.c_contiguous
    assert query.strides[1] == query.itemsize and key.strides[1] == key.itemsize
    if T == 0:
        return
    rc = lib().pao_rotary_embedding_neox(
        _ptr(positions), _ptr(query), _ptr(key), _ptr(cos_sin_cache), T, query.shape[1] // head_size, head_size,
        cos_sin_cache.shape[1], _elem_stride(query, 0), _elem_stride(key, 0), DTYPE_CODE[dtype])
    if rc != 0:
        raise RuntimeError(f"pa_oracle rotary_embedding_neox failed with code {rc}")


def rms_norm(input: np.ndarray, weight: np.ndarray, epsilon: float, dtype: str) -> np.ndarray:
    assert input.flags.c_contiguous and weight.flags.c_contiguous
    out = np.zeros_like(input)
    if input.size:
        rc = lib().pao_rms_norm(_ptr(out), _ptr(input), _ptr(weight), float(epsilon), input.shape[0],
                                input.shape[1], DTYPE_CODE[dtype])
        if rc != 0:
            raise RuntimeError(f"pa_oracle rms_norm failed with code {rc}")
    return out


def silu_and_mul(input: np.ndarray, dtype: str) -> np.ndarray:
    assert input.flags.c_contiguous
    T, d2 = input.shape
    out = np.zeros((T, d2 // 2), dtype=input.dtype)
    if out.size:
        rc = lib().pao_silu_and_mul(_ptr(out), _ptr(input), T, d2 // 2, DTYPE_CODE[dtype])
        if rc != 0:
            raise RuntimeError(f"pa_oracle silu_and_mul failed with code {rc}")
    return out


def multi_query_kv_attention(query: np.ndarray, key: np.ndarray, value: np.ndarray,
                             cu_seqlens: Sequence[int], scale: float, dtype: str) -> np.ndarray:
    """Causal attention inside each prompt; query / key / value are [T, H, D]
    (strided views are fine), the result is dense [T, H, D]."""
    T, H, D = query.shape
    for a in (query, key, value):
        _check_inner_dense(a)
    cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
    assert cu[0] == 0 and cu[-1] <= T
    out = np.zeros((T, H, D), dtype=query.dtype)
    if len(cu) > 1 and cu[-1] > 0:
        rc = lib().pao_multi_query_kv_attention(
            _ptr(out), _ptr(query), _ptr(key), _ptr(value), _ptr(cu), len(cu) - 1, H, D, float(scale),
            _elem_stride(query, 0), _elem_stride(key, 0), _elem_stride(value, 0), DTYPE_CODE[dtype])
        if rc != 0:
            raise RuntimeError(f"pa_oracle multi_query_kv_attention failed with code {rc}")
    return out
