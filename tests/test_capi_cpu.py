"""CPU tests of the boundary: the shared library loads, exports every symbol
the public header declares, and rejects bad arguments before touching the GPU.
No kernel is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    names = []
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(inc, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_expected_entry_points(pkg):
    declared = _declared_functions()
    assert set(declared) == set(pkg._capi.EXPORTS), declared


def test_library_exports_every_declared_symbol(pkg):
    lib = ctypes.CDLL(pkg._capi.LIB_PATH)
    for name in _declared_functions():
        assert getattr(lib, name) is not None, name
    assert pkg._capi.lib.pa_abi_version() == pkg._capi.PA_ABI_VERSION


def test_operator_modules_export_the_reference_names(pkg):
    # csrc/attention.cpp:14-19 and csrc/cache.cpp:30-47
    assert callable(pkg.attention_ops.single_query_cached_kv_attention)
    for name in ("swap_blocks", "copy_blocks", "reshape_and_cache", "gather_cached_kv"):
        assert callable(getattr(pkg.cache_ops, name)), name


def test_argument_validation_needs_no_gpu(pkg):
    c = pkg._capi
    lib = c.lib
    one = ctypes.c_void_p(16)  # non-null, 16-byte aligned dummy; never dereferenced on these paths

    def decode(**kw):
        a = dict(S=2, H=2, D=64, B=16, mb=1, mc=16, qs=128, dtype=c.PA_DTYPE_F16,
                 out=one, q=one, kc=one, vc=one, bt=one, cl=one)
        a.update(kw)
        return lib.pa_single_query_cached_kv_attention(
            a["out"], a["q"], a["kc"], a["vc"], 1.0, a["bt"], a["cl"], a["S"], a["H"], a["D"], a["B"],
            a["mb"], a["mc"], a["qs"], a["dtype"], None)

    assert decode(dtype=7) == c.PA_ERR_UNSUPPORTED and "data type" in c.last_error()
    assert decode(D=100) == c.PA_ERR_UNSUPPORTED and "head size: 100" in c.last_error()
    assert decode(B=64) == c.PA_ERR_UNSUPPORTED and "block size: 64" in c.last_error()
    assert decode(S=0) == c.PA_OK and c.last_error() == ""
    assert decode(H=0) == c.PA_OK
    assert decode(S=-1) == c.PA_ERR_INVALID_ARGUMENT
    assert decode(q=None) == c.PA_ERR_INVALID_ARGUMENT and "null" in c.last_error()
    assert decode(kc=ctypes.c_void_p(8)) == c.PA_ERR_INVALID_ARGUMENT and "aligned" in c.last_error()
    assert decode(mb=1, mc=17) == c.PA_ERR_INVALID_ARGUMENT and "cannot hold" in c.last_error()

    def token(fn, **kw):
        a = dict(T=2, H=2, D=64, B=16, dtype=c.PA_DTYPE_F16, k=one, v=one, kc=one, vc=one, sm=one)
        a.update(kw)
        return fn(a["k"], a["v"], a["kc"], a["vc"], a["sm"], a["T"], a["H"], a["D"], a["B"], 128, 128,
                  a["dtype"], None)

    for fn in (lib.pa_reshape_and_cache, lib.pa_gather_cached_kv):
        assert token(fn, T=0) == c.PA_OK
        assert token(fn, dtype=9) == c.PA_ERR_UNSUPPORTED
        assert token(fn, D=12) == c.PA_ERR_INVALID_ARGUMENT and "multiple of x" in c.last_error()
        assert token(fn, sm=None) == c.PA_ERR_INVALID_ARGUMENT
        assert token(fn, B=0) == c.PA_ERR_INVALID_ARGUMENT

    assert lib.pa_copy_blocks(None, None, 0, None, 0, 0, None) == c.PA_OK
    assert lib.pa_copy_blocks(None, None, 2, None, 3, 64, None) == c.PA_ERR_INVALID_ARGUMENT
    pairs = (ctypes.c_int64 * 2)(0, -1)
    ptrs = (ctypes.c_void_p * 1)(16)
    assert lib.pa_copy_blocks(ptrs, ptrs, 1, pairs, 1, 64, None) == c.PA_ERR_INVALID_ARGUMENT
    assert lib.pa_swap_blocks(None, None, None, 0, 64, 0, None) == c.PA_OK
    ok_pairs = (ctypes.c_int64 * 2)(0, 1)
    assert lib.pa_swap_blocks(one, one, ok_pairs, 1, 64, 5, None) == c.PA_ERR_INVALID_ARGUMENT
    assert "device combination" in c.last_error()


def test_operator_modules_reject_cpu_tensors_and_bad_shapes(pkg):
    import torch
    q = torch.zeros(2, 2, 64, dtype=torch.float16)
    kc = torch.zeros(4, 2, 8, 16, 8, dtype=torch.float16)
    vc = torch.zeros(4, 2, 64, 16, dtype=torch.float16)
    bt = torch.zeros(2, 1, dtype=torch.int32)
    cl = torch.ones(2, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="must be on the GPU"):
        pkg.attention_ops.single_query_cached_kv_attention(q.clone(), q, kc, vc, 1.0, bt, cl, 16, 1)
    sm = torch.zeros(2, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="must be on the GPU"):
        pkg.cache_ops.reshape_and_cache(q, q, kc, vc, sm)
    with pytest.raises(RuntimeError, match="must be on the GPU"):
        pkg.cache_ops.gather_cached_kv(q, q, kc, vc, sm)
    # cpu <-> cpu is the reference's "Invalid device combination" (cache_kernels.cu:25-27)
    with pytest.raises(RuntimeError, match="Invalid device combination"):
        pkg.cache_ops.swap_blocks(kc, kc.clone(), {0: 1})
    # empty layer list returns quietly (cache_kernels.cu:88-90)
    pkg.cache_ops.copy_blocks([], [], {0: [1]})
    with pytest.raises(RuntimeError, match="must be on the GPU"):
        pkg.cache_ops.copy_blocks([kc], [vc], {0: [1]})


def test_communicator_and_append_entry_validate_arguments_without_a_gpu(pkg):
    c = pkg._capi
    lib = c.lib
    comm = ctypes.c_void_p()
    blob = (ctypes.c_ubyte * c.COMM_HANDLE_BYTES)()
    assert lib.pa_comm_create(3, 2, 1024, ctypes.byref(comm), blob) == c.PA_ERR_INVALID_ARGUMENT
    assert "rank 3 of 2" in c.last_error()
    assert lib.pa_comm_create(0, 9, 1024, ctypes.byref(comm), blob) == c.PA_ERR_INVALID_ARGUMENT
    assert lib.pa_comm_create(0, 2, 0, ctypes.byref(comm), blob) == c.PA_ERR_INVALID_ARGUMENT
    assert "max_bytes" in c.last_error()
    assert lib.pa_comm_all_reduce(None, None, 8, c.PA_DTYPE_F16, 0, None) == c.PA_ERR_INVALID_ARGUMENT
    assert lib.pa_comm_connect(None, None) == c.PA_ERR_INVALID_ARGUMENT
    assert lib.pa_comm_destroy(None) == c.PA_OK
    one = ctypes.c_void_p(16)
    # the new token rows must be at least num_heads x head_size elements apart
    rc = lib.pa_append_single_query_cached_kv_attention(
        one, one, one, one, one, one, 1.0, one, one, 2, 2, 64, 16, 1, 16, 128, 64, 128, c.PA_DTYPE_F16, None, 0, 0, -1, None)
    assert rc == c.PA_ERR_INVALID_ARGUMENT and "shorter than a token row" in c.last_error()
    rc = lib.pa_append_single_query_cached_kv_attention(
        one, one, None, one, one, one, 1.0, one, one, 2, 2, 64, 16, 1, 16, 128, 128, 128, c.PA_DTYPE_F16, None, 0, 0, -1, None)
    assert rc == c.PA_ERR_INVALID_ARGUMENT and "null key / value" in c.last_error()
