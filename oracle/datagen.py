"""Deterministic synthetic inputs for the PagedAttention decode path.

TEST INFRASTRUCTURE (used by tests/, oracle/gen_golden.py and smoke()).  The
generator is a counter-based hash (splitmix64 finaliser over an index), so
the same (seed, shape) gives the same bits on every numpy version and on
every machine; golden fixtures therefore only need to store a checksum of the
inputs plus the expected outputs.

Input conventions follow the reference's kernel tests and its worker:
  * q is slice 0 of a fused qkv tensor [S, 3, H, D], so its row stride is
    3*H*D (tests/kernels/test_attention.py:168-171);
  * key_cache [NB, H, D/x, B, x], value_cache [NB, H, D, B], x = 16/elem size
    (vllm/worker/cache_engine.py:48-63);
  * block_tables int32, right-padded with 0 (vllm/worker/worker.py:219-222);
  * context_len counts the token whose K/V was just written
    (vllm/worker/worker.py:192).
"""
from __future__ import annotations

import hashlib
import json
from dataclasses import asdict, dataclass, field
from typing import Dict, List, Optional

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
DTYPES = ("float32", "float16", "bfloat16")
DTYPE_CODE = {"float32": 0, "float16": 1, "bfloat16": 2}
ELEM_SIZE = {"float32": 4, "float16": 2, "bfloat16": 2}


def _mix(idx: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64 finaliser of (idx + seed * golden ratio)."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(n: int, seed: int) -> np.ndarray:
    """n doubles in [0, 1)."""
    z = _mix(np.arange(n, dtype=np.uint64), seed)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(n: int, seed: int, lo: float, hi: float) -> np.ndarray:
    return lo + (hi - lo) * uniform01(n, seed)


def normal(n: int, seed: int, std: float = 1.0) -> np.ndarray:
    """Box-Muller on two hashed uniform streams."""
    u1 = uniform01(n, seed * 2 + 1)
    u2 = uniform01(n, seed * 2 + 2)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return std * r * np.cos(2.0 * np.pi * u2)


def randint(n: int, seed: int, lo: int, hi: int) -> np.ndarray:
    """n integers in [lo, hi] (inclusive, like random.randint)."""
    z = _mix(np.arange(n, dtype=np.uint64), seed)
    return (lo + (z % np.uint64(hi - lo + 1)).astype(np.int64)).astype(np.int64)


def permutation(n: int, seed: int) -> np.ndarray:
    """A permutation of arange(n): argsort of a hashed key (ties impossible
    in practice for 64-bit keys; stable sort keeps it deterministic)."""
    return np.argsort(_mix(np.arange(n, dtype=np.uint64), seed), kind="stable").astype(np.int64)


# ---- storage types -------------------------------------------------------

def float_to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """float32 -> bfloat16 bit patterns, round to nearest even."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    lsb = (u >> np.uint64(16)) & np.uint64(1)
    r = (u + np.uint64(0x7FFF) + lsb) >> np.uint64(16)
    return r.astype(np.uint16)


def bf16_bits_to_float(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def to_storage(a: np.ndarray, dtype: str) -> np.ndarray:
    """Round float64 values to the storage type.  bfloat16 is held as uint16
    bit patterns (numpy has no bfloat16)."""
    if dtype == "float32":
        return a.astype(np.float32)
    if dtype == "float16":
        return a.astype(np.float32).astype(np.float16)
    if dtype == "bfloat16":
        return float_to_bf16_bits(a.astype(np.float32))
    raise ValueError(dtype)


def storage_to_float32(a: np.ndarray, dtype: str) -> np.ndarray:
    if dtype == "bfloat16":
        return bf16_bits_to_float(a)
    return a.astype(np.float32)


def storage_to_torch(a: np.ndarray, dtype: str):
    import torch
    t = torch.from_numpy(a)
    if dtype == "bfloat16":
        t = t.view(torch.bfloat16)
    return t


def torch_to_storage(t, dtype: str) -> np.ndarray:
    import torch
    t = t.detach().cpu().contiguous()
    if dtype == "bfloat16":
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    return t.numpy().copy()


# ---- decode-attention cases ----------------------------------------------

@dataclass
class DecodeCase:
    name: str
    dtype: str = "float16"
    num_seqs: int = 5
    num_heads: int = 3
    head_size: int = 64
    block_size: int = 16
    num_blocks: int = 32
    # explicit context lengths, or None -> randint(1, max_context) per seq
    context_lens: Optional[List[int]] = None
    max_context: int = 64
    # "uniform1e-3": the reference test's distribution
    # (tests/kernels/test_attention.py:170,177,181);  "normal": N(0, std)
    dist: str = "uniform1e-3"
    std: float = 0.5
    # "replacement": physical blocks sampled with replacement, as the
    # reference test does (:190-194); "permutation": all distinct;
    # "forked": groups of `fork` consecutive sequences sampled from one prompt
    # (parallel sampling, n = fork): the group maps the SAME physical blocks for
    # every full block of its prompt and distinct ones from there on — what the
    # block manager's fork + copy-on-write leaves behind
    # (vllm/core/block_manager.py:111-135)
    tables: str = "replacement"
    fork: int = 3
    # extra zero-padded table columns beyond ceil(max_ctx / B)
    table_pad: int = 0
    # None -> 1/sqrt(head_size), as the reference test (:197)
    scale: Optional[float] = None
    seed: int = 0

    def to_json(self) -> str:
        return json.dumps(asdict(self), sort_keys=True)

    @staticmethod
    def from_json(s: str) -> "DecodeCase":
        return DecodeCase(**json.loads(s))


def _values(n: int, seed: int, case: DecodeCase) -> np.ndarray:
    if case.dist == "uniform1e-3":
        return uniform(n, seed, -1e-3, 1e-3)
    if case.dist == "normal":
        return normal(n, seed, case.std)
    raise ValueError(case.dist)


def make_decode_inputs(case: DecodeCase) -> Dict[str, object]:
    """Builds numpy inputs for one decode-attention call.

    Returns a dict with: qkv (storage array [S,3,H,D]; query = qkv[:,0]),
    key_cache, value_cache (storage arrays), block_tables, context_lens
    (int32), scale, max_context_len, q_stride.
    """
    S, H, D, B, NB = case.num_seqs, case.num_heads, case.head_size, case.block_size, case.num_blocks
    es = ELEM_SIZE[case.dtype]
    x = 16 // es
    assert D % x == 0
    base = case.seed * 1000
    qkv = to_storage(_values(S * 3 * H * D, base + 1, case), case.dtype).reshape(S, 3, H, D)
    kc = to_storage(_values(NB * H * D * B, base + 2, case), case.dtype).reshape(NB, H, D // x, B, x)
    vc = to_storage(_values(NB * H * D * B, base + 3, case), case.dtype).reshape(NB, H, D, B)
    if case.context_lens is not None:
        ctx = np.asarray(case.context_lens, dtype=np.int64)
        assert ctx.shape == (S,)
    else:
        ctx = randint(S, base + 4, 1, case.max_context)
    max_ctx = int(ctx.max()) if S > 0 else 0
    nb_max = (max_ctx + B - 1) // B
    cols = nb_max + case.table_pad
    tables = np.zeros((S, max(cols, 1)), dtype=np.int32)
    if case.tables == "replacement":
        r = randint(S * max(nb_max, 1), base + 5, 0, NB - 1).reshape(S, max(nb_max, 1))
        for s in range(S):
            n = (int(ctx[s]) + B - 1) // B
            tables[s, :n] = r[s, :n]
    elif case.tables == "permutation":
        need = int(sum((int(c) + B - 1) // B for c in ctx))
        assert need <= NB, "permutation tables need num_blocks >= total logical blocks"
        perm = permutation(NB, base + 5)
        pos = 0
        for s in range(S):
            n = (int(ctx[s]) + B - 1) // B
            tables[s, :n] = perm[pos:pos + n]
            pos += n
    elif case.tables == "forked":
        F = max(int(case.fork), 1)
        perm = permutation(NB, base + 5)
        gen = randint((S + F - 1) // F, base + 6, 1, 2 * B)  # tokens generated since the fork, per group
        pos = 0
        for g0 in range(0, S, F):
            members = range(g0, min(g0 + F, S))
            prompt_len = max(int(min(ctx[s] for s in members)) - int(gen[g0 // F]), 0)
            shared = prompt_len // B  # full prompt blocks stay shared; a partial last one was copied on write
            assert pos + shared <= NB, "forked tables: num_blocks too small"
            shared_blocks = perm[pos:pos + shared]
            pos += shared
            for s in members:
                n = (int(ctx[s]) + B - 1) // B
                own = n - shared
                assert own >= 0 and pos + own <= NB, "forked tables: num_blocks too small"
                tables[s, :shared] = shared_blocks
                tables[s, shared:n] = perm[pos:pos + own]
                pos += own
    else:
        raise ValueError(case.tables)
    scale = float(case.scale) if case.scale is not None else float(1.0 / (D ** 0.5))
    return {
        "qkv": qkv,
        "key_cache": kc,
        "value_cache": vc,
        "block_tables": tables,
        "context_lens": ctx.astype(np.int32),
        "scale": scale,
        "max_context_len": max_ctx,
        "q_stride": 3 * H * D,
    }


def checksum(inputs: Dict[str, object]) -> str:
    h = hashlib.sha256()
    for k in ("qkv", "key_cache", "value_cache", "block_tables", "context_lens"):
        a = np.ascontiguousarray(inputs[k])
        h.update(k.encode())
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    h.update(repr(float(inputs["scale"])).encode())
    return h.hexdigest()


# ---- cache-op cases -------------------------------------------------------

@dataclass
class CacheCase:
    name: str
    dtype: str = "float16"
    num_tokens: int = 3
    num_heads: int = 2
    head_size: int = 16
    block_size: int = 8
    num_blocks: int = 2
    num_layers: int = 1
    num_mappings: int = 0
    seed: int = 0
    extra: Dict[str, int] = field(default_factory=dict)

    def to_json(self) -> str:
        return json.dumps(asdict(self), sort_keys=True)

    @staticmethod
    def from_json(s: str) -> "CacheCase":
        return CacheCase(**json.loads(s))


def make_cache_inputs(case: CacheCase) -> Dict[str, object]:
    """Inputs in the shape of tests/kernels/test_cache.py: randn caches (one
    pair per layer), a qkv tensor whose slices 1 and 2 are key and value,
    distinct random slots, and disjoint src/dst block lists."""
    T, H, D, B, NB, L = (case.num_tokens, case.num_heads, case.head_size, case.block_size,
                         case.num_blocks, case.num_layers)
    es = ELEM_SIZE[case.dtype]
    x = 16 // es
    assert D % x == 0
    base = case.seed * 1000 + 500
    qkv = to_storage(normal(max(T, 1) * 3 * H * D, base + 1), case.dtype).reshape(max(T, 1), 3, H, D)[:T]
    kcs, vcs = [], []
    for l in range(L):
        kcs.append(to_storage(normal(NB * H * D * B, base + 10 + 2 * l), case.dtype).reshape(NB, H, D // x, B, x))
        vcs.append(to_storage(normal(NB * H * D * B, base + 11 + 2 * l), case.dtype).reshape(NB, H, D, B))
    slots = permutation(NB * B, base + 2)[:T].astype(np.int32)
    perm = permutation(NB, base + 3)
    M = case.num_mappings
    assert 2 * M <= NB
    src = perm[:M]
    dst = perm[M:2 * M]
    return {"qkv": qkv, "key_caches": kcs, "value_caches": vcs, "slot_mapping": slots,
            "src_blocks": src.astype(np.int64), "dst_blocks": dst.astype(np.int64)}


# ---- element-wise operator cases (rotary embedding, RMS norm, SiLU-and-mul) --

@dataclass
class ElementwiseCase:
    name: str
    op: str                 # "rotary" | "rms_norm" | "silu_and_mul"
    dtype: str = "float16"
    num_tokens: int = 7
    num_heads: int = 3      # rotary
    head_size: int = 64     # rotary
    rot_dim: int = 64       # rotary
    max_position: int = 64  # rotary
    hidden: int = 64        # rms_norm: hidden size; silu_and_mul: d
    weight: str = "uniform1e-3"  # rms_norm: the reference test's weights, or "normal1" = N(1, 0.1)
    seed: int = 0

    def to_json(self) -> str:
        return json.dumps(asdict(self), sort_keys=True)


def cos_sin_cache(max_position: int, rot_dim: int, dtype: str, base: float = 10000.0) -> np.ndarray:
    """[max_position, rot_dim] = cos | sin of position * base^(-2i/rot_dim), the
    table PagedAttentionWithRoPE registers
    (vllm/model_executor/layers/attention.py:173-186)."""
    inv_freq = 1.0 / (base ** (np.arange(0, rot_dim, 2, dtype=np.float64) / rot_dim))
    freqs = np.arange(max_position, dtype=np.float64)[:, None] * inv_freq[None, :]
    return to_storage(np.concatenate([np.cos(freqs), np.sin(freqs)], axis=-1), dtype)


def make_elementwise_inputs(case: ElementwiseCase) -> Dict[str, np.ndarray]:
    base = case.seed * 1000 + 700
    T = case.num_tokens
    if case.op == "rotary":
        n = case.num_heads * case.head_size
        return {
            "positions": randint(T, base + 1, 0, case.max_position - 1),
            "query": to_storage(normal(T * n, base + 2), case.dtype).reshape(T, n),
            "key": to_storage(normal(T * n, base + 3), case.dtype).reshape(T, n),
            "cos_sin_cache": cos_sin_cache(case.max_position, case.rot_dim, case.dtype),
        }
    if case.op == "rms_norm":
        if case.weight == "uniform1e-3":
            w = uniform(case.hidden, base + 2, -1e-3, 1e-3)
        else:
            w = 1.0 + normal(case.hidden, base + 2, 0.1)
        return {
            "input": to_storage(normal(T * case.hidden, base + 1), case.dtype).reshape(T, case.hidden),
            "weight": to_storage(w, case.dtype),
        }
    if case.op == "silu_and_mul":
        return {"input": to_storage(normal(T * 2 * case.hidden, base + 1, 2.0), case.dtype).reshape(T, 2 * case.hidden)}
    raise ValueError(case.op)


def checksum_arrays(arrays: Dict[str, np.ndarray]) -> str:
    h = hashlib.sha256()
    for k in sorted(arrays):
        a = np.ascontiguousarray(arrays[k])
        h.update(k.encode())
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def elementwise_cases() -> List[ElementwiseCase]:
    cases: List[ElementwiseCase] = []
    seed = 300
    for dtype in DTYPES:
        # tests/kernels/test_pos_encoding.py:136-146: 8 head sizes, rot_dim = head size, 5 heads
        for hs in (32, 64, 80, 96, 128, 160, 192, 256):
            cases.append(ElementwiseCase(name=f"rotary_{dtype}_h{hs}", op="rotary", dtype=dtype, num_tokens=19,
                                         num_heads=5, head_size=hs, rot_dim=hs, max_position=64, seed=seed))
            seed += 1
        # partial rotary (GPT-NeoX rotary_pct, vllm/model_executor/models/gpt_neox.py:64) and a
        # rot_dim whose half is not a multiple of a 16-byte vector
        for hs, rd in ((96, 24), (128, 32), (64, 20)):
            cases.append(ElementwiseCase(name=f"rotary_{dtype}_h{hs}_r{rd}", op="rotary", dtype=dtype, num_tokens=11,
                                         num_heads=3, head_size=hs, rot_dim=rd, max_position=64, seed=seed))
            seed += 1
        # tests/kernels/test_layernorm.py:46-57
        for hidden in (13, 64, 1024, 5120):
            for w in ("uniform1e-3", "normal1"):
                cases.append(ElementwiseCase(name=f"rms_{dtype}_{hidden}_{w}", op="rms_norm", dtype=dtype,
                                             num_tokens=7, hidden=hidden, weight=w, seed=seed))
                seed += 1
        # tests/kernels/test_activation.py:27-33 (smaller d, plus one that is not a multiple of 8)
        for d in (512, 1001, 5120):
            cases.append(ElementwiseCase(name=f"silu_{dtype}_{d}", op="silu_and_mul", dtype=dtype, num_tokens=7,
                                         hidden=d, seed=seed))
            seed += 1
    return cases


# ---- prompt attention cases -------------------------------------------------

@dataclass
class PromptCase:
    name: str
    dtype: str = "float16"
    seq_lens: Optional[List[int]] = None
    num_heads: int = 2
    head_size: int = 128
    dist: str = "normal"     # "normal" | "uniform1e-3" (tests/kernels/test_attention.py:241)
    pad_tokens: int = 0      # padding rows after the last prompt (attention.py:83-87)
    seed: int = 0

    def to_json(self) -> str:
        return json.dumps(asdict(self), sort_keys=True)


def make_prompt_inputs(case: PromptCase) -> Dict[str, np.ndarray]:
    """qkv [T, 3, H, D] in storage form (query, key, value are its strided
    slices, as in tests/kernels/test_attention.py:238-242) and cu_seqlens."""
    T = int(sum(case.seq_lens)) + case.pad_tokens
    n = T * 3 * case.num_heads * case.head_size
    base = case.seed * 1000 + 900
    vals = normal(n, base + 1) if case.dist == "normal" else uniform(n, base + 1, -1e-3, 1e-3)
    qkv = to_storage(vals, case.dtype).reshape(T, 3, case.num_heads, case.head_size)
    cu = np.concatenate([[0], np.cumsum(case.seq_lens)]).astype(np.int32)
    return {"qkv": qkv, "cu_seqlens": cu}


def prompt_cases() -> List[PromptCase]:
    cases: List[PromptCase] = []
    seed = 500
    for dtype in DTYPES:
        # the reference test's grid (tests/kernels/test_attention.py:308-320) at small lengths
        for D in (64, 80, 96, 128):
            cases.append(PromptCase(name=f"grid_{dtype}_d{D}", dtype=dtype, seq_lens=[1, 7, 33, 40, 70],
                                    num_heads=2, head_size=D, dist="normal", seed=seed))
            seed += 1
        cases.append(PromptCase(name=f"reftest_dist_{dtype}", dtype=dtype, seq_lens=[3, 64, 65],
                                num_heads=3, head_size=128, dist="uniform1e-3", seed=seed))
        seed += 1
        # lengths around the kernel's tile edges, one long prompt, padding rows behind
        cases.append(PromptCase(name=f"edges_{dtype}", dtype=dtype, seq_lens=[128, 1, 129, 31, 32, 300],
                                num_heads=1, head_size=128 if dtype != "float32" else 64, dist="normal",
                                pad_tokens=5, seed=seed))
        seed += 1
    return cases


# ---- scheduler output for the input builder --------------------------------------

class SeqData:
    """The three accessors of vllm.sequence.SequenceData that
    Worker._prepare_inputs uses (vllm/worker/worker.py:151,189,192)."""

    def __init__(self, tokens):
        self.tokens = list(tokens)

    def get_len(self):
        return len(self.tokens)

    def get_token_ids(self):
        return self.tokens

    def get_last_token_id(self):
        return self.tokens[-1]


def make_input_batch(seed: int, num_prompt_groups: int, num_gen_groups: int, block_size: int,
                     profiling: bool = False) -> list:
    """A scheduler output in the shape of List[SequenceGroupMetadata]
    (vllm/sequence.py): prompt groups of one or two sequences sharing a prompt
    and its block table, generation groups of one to three sequences with
    their own tables, prompt and generation groups interleaved.  profiling:
    prompt groups carry block_tables=None, as in the memory-profiling run
    (vllm/worker/worker.py:160-164)."""
    from types import SimpleNamespace
    groups = []
    next_id, next_block = 0, 1
    r = randint(4 * (num_prompt_groups + num_gen_groups) + 8, seed, 1, 70)
    k = 0
    for _ in range(num_prompt_groups):
        n = int(r[k]); k += 1
        ids = [next_id, next_id + 1] if r[k] % 3 == 0 else [next_id]
        k += 1
        next_id += len(ids)
        toks = [int(t) for t in randint(n, seed + next_id, 0, 31999)]
        nb = (n + block_size - 1) // block_size
        table = list(range(next_block, next_block + nb))[::-1]
        next_block += nb
        groups.append(SimpleNamespace(
            request_id=str(next_id), is_prompt=True, seq_data={i: SeqData(toks) for i in ids},
            sampling_params=SimpleNamespace(n=len(ids)),
            block_tables=None if profiling else {i: table for i in ids}))
    for _ in range(num_gen_groups):
        nseq = 1 + int(r[k]) % 3; k += 1
        seq_data, tables = {}, {}
        for _s in range(nseq):
            n = int(r[k % len(r)]) + _s; k += 1
            toks = [int(t) for t in randint(n, seed + 100 + next_id, 0, 31999)]
            nb = (n + block_size - 1) // block_size
            tables[next_id] = list(range(next_block, next_block + nb))
            next_block += nb
            seq_data[next_id] = SeqData(toks)
            next_id += 1
        groups.append(SimpleNamespace(request_id=str(next_id), is_prompt=False, seq_data=seq_data,
                                      sampling_params=SimpleNamespace(n=nseq), block_tables=tables))
    # the scheduler may interleave prompt and generation groups
    order = permutation(len(groups), seed + 7)
    return [groups[i] for i in order]


def input_batch_cases() -> List[dict]:
    """(seed, prompt groups, generation groups, block size, profiling) of the input-builder fixtures."""
    cases = []
    for block_size in (8, 16, 32):
        for shape in ((3, 5), (0, 7), (4, 0), (1, 1), (0, 0), (9, 40)):
            cases.append({"seed": 11 + block_size, "prompts": shape[0], "gens": shape[1],
                          "block_size": block_size, "profiling": False})
    cases.append({"seed": 5, "prompts": 3, "gens": 0, "block_size": 16, "profiling": True})
    return cases
