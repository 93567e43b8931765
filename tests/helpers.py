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
REF_ATOL, REF_RTOL = 1e-3, 1e-5
TIGHT = {"float32": (2e-5, 2e-5), "float16": (2e-5, 1.0e-3), "bfloat16": (2e-5, 8.0e-3)}


def load_decode_golden() -> Tuple[Dict[str, dict], "np.lib.npyio.NpzFile"]:
    with open(os.path.join(GOLDEN_DIR, "decode_golden.json")) as f:
        meta = json.load(f)
    data = np.load(os.path.join(GOLDEN_DIR, "decode_golden.npz"), allow_pickle=False)
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


def assert_close(got_f32: np.ndarray, want_f32: np.ndarray, atol: float, rtol: float, what: str) -> None:
    got_f32 = np.asarray(got_f32, dtype=np.float64)
    want_f32 = np.asarray(want_f32, dtype=np.float64)
    assert got_f32.shape == want_f32.shape, (what, got_f32.shape, want_f32.shape)
    assert np.all(np.isfinite(got_f32)), f"{what}: non-finite values in the result"
    err = np.abs(got_f32 - want_f32)
    bound = atol + rtol * np.abs(want_f32)
    if not np.all(err <= bound):
        i = np.unravel_index(np.argmax(err - bound), err.shape)
        raise AssertionError(
            f"{what}: max violation at {i}: got {got_f32[i]!r}, want {want_f32[i]!r}, "
            f"|err| {err[i]:.3e} > {bound[i]:.3e} (atol {atol}, rtol {rtol}); max |err| {err.max():.3e}")


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
