"""Shared helpers of the parity tests."""
from __future__ import annotations

import json
import os
from typing import Dict, Tuple

import numpy as np

from oracle import datagen
from oracle.datagen import CacheCase, DecodeCase

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Tolerances.
#  * REF_ATOL / REF_RTOL: the reference test's own criterion
#    (tests/kernels/test_attention.py:224).
#  * against the fp32 evaluation of the reference formula the only error left
#    is fp32 accumulation order plus the final rounding to the storage type:
#    half an ulp of the output (2^-11 relative for fp16, 2^-8 for bf16).
#  * bfloat16 decode attention rounds the softmax weights to bfloat16 before
#    the weighted sum of V, as the reference kernel does
#    (csrc/attention/attention_kernels.cu:236-254: logits converted to the
#    storage type, then dotted with V).  Each weight is off by at most 2^-8
#    relative (half an ulp of 8 significant bits), so an output element is off
#    by at most 2^-8 * sum_i p_i |v_i| / sum_i p_i on top of the above: decode_atol()
#    evaluates that bound per element with the oracle run on |V|.
REF_ATOL, REF_RTOL = 1e-3, 1e-5
TIGHT = {"float32": (2e-5, 2e-5), "float16": (2e-5, 1.0e-3), "bfloat16": (2e-5, 8.0e-3)}


def decode_atol(inp: dict, dtype: str):
    """Absolute tolerance of a decode-attention result against the fp32
    evaluation: TIGHT's scalar, plus for bfloat16 the per-element bound of the
    weights' rounding (see above)."""
    atol = TIGHT[dtype][0]
    if dtype != "bfloat16":
        return atol
    from oracle import pa_oracle
    v_abs = (np.asarray(inp["value_cache"]).view(np.uint16) & np.uint16(0x7FFF)).view(inp["value_cache"].dtype)
    mass = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], v_abs, inp["scale"], inp["block_tables"], inp["context_lens"], dtype), dtype)
    return atol + 2.0 ** -8 * 1.05 * np.abs(mass.astype(np.float64))


def load_decode_golden() -> Tuple[Dict[str, dict], Dict[str, np.ndarray]]:
    """The reference-generated decode fixtures: the reference test's grid and
    edge cases (decode_golden) and the per-GPU shapes of the tensor-parallel
    configurations, with forked block tables (decode_golden_shards)."""
    meta, data = {}, {}
    for stem in ("decode_golden", "decode_golden_shards"):
        with open(os.path.join(GOLDEN_DIR, stem + ".json")) as f:
            meta.update(json.load(f))
        with np.load(os.path.join(GOLDEN_DIR, stem + ".npz"), allow_pickle=False) as z:
            data.update({k: z[k] for k in z.files})
    return meta, data


def load_cache_golden() -> Tuple[Dict[str, dict], "np.lib.npyio.NpzFile"]:
    with open(os.path.join(GOLDEN_DIR, "cache_golden.json")) as f:
        meta = json.load(f)
    data = np.load(os.path.join(GOLDEN_DIR, "cache_golden_small.npz"), allow_pickle=False)
    return meta, data


def decode_case_from_meta(entry: dict) -> DecodeCase:
    return DecodeCase(**entry["case"])


def cache_case_from_meta(entry: dict) -> CacheCase:
    return CacheCase(**entry["case"])


def assert_close(got_f32: np.ndarray, want_f32: np.ndarray, atol, rtol: float, what: str) -> None:
    got_f32 = np.asarray(got_f32, dtype=np.float64)
    want_f32 = np.asarray(want_f32, dtype=np.float64)
    assert got_f32.shape == want_f32.shape, (what, got_f32.shape, want_f32.shape)
    assert np.all(np.isfinite(got_f32)), f"{what}: non-finite values in the result"
    err = np.abs(got_f32 - want_f32)
    bound = atol + rtol * np.abs(want_f32)
    if not np.all(err <= bound):
        i = np.unravel_index(np.argmax(err - bound), err.shape)
        atol_i = float(np.broadcast_to(atol, err.shape)[i])
        raise AssertionError(
            f"{what}: max violation at {i}: got {got_f32[i]!r}, want {want_f32[i]!r}, "
            f"|err| {err[i]:.3e} > {bound[i]:.3e} (atol {atol_i:.3e}, rtol {rtol}); max |err| {err.max():.3e}")


def decode_inputs_to_device(inp: dict, dtype: str, device):
    """numpy storage arrays -> torch tensors on `device` in the reference's
    calling convention (query is a strided slice of the fused qkv tensor)."""
    import torch
    qkv = datagen.storage_to_torch(inp["qkv"], dtype).to(device)
    query = qkv[:, 0]
    kc = datagen.storage_to_torch(inp["key_cache"], dtype).to(device)
    vc = datagen.storage_to_torch(inp["value_cache"], dtype).to(device)
    bt = torch.from_numpy(inp["block_tables"]).to(device)
    cl = torch.from_numpy(inp["context_lens"]).to(device)
    return qkv, query, kc, vc, bt, cl


def out_to_f32(t, dtype: str) -> np.ndarray:
    return t.detach().float().cpu().numpy()
