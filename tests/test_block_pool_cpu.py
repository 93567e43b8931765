"""The throughput harness's block bookkeeping (tools/bench_throughput.py: BlockPool, Sequence.fork), on the CPU:
reference counts of forked sequences, copy-on-write accounting and swap mappings in which a block shared by the
sequences of a group moves once (vllm/core/block_manager.py:111-135, 196-224)."""
import importlib.util
import os


def _harness():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench_throughput.py")
    spec = importlib.util.spec_from_file_location("bench_throughput_cpu", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fork_share_copy_on_write_and_release():
    mod = _harness()
    pool = mod.BlockPool(8, 4)
    parent = mod.Sequence(list(range(20)), 5)          # 20 tokens, block size 16: two blocks, the second shared later
    parent.table = [pool.alloc(), pool.alloc()]
    kids = [parent.fork(), parent.fork()]
    for k in kids:
        pool.share(k.table)
    assert all(pool.ref[b] == 3 for b in parent.table) and len(pool.free_gpu) == 6
    assert kids[0].tokens == parent.tokens and kids[0].tokens is not parent.tokens and kids[0].prompt_len == 20
    # first append of a child into the shared last block: a private copy, the source loses one reference
    src = kids[0].table[-1]
    pool.ref[src] -= 1
    kids[0].table[-1] = pool.alloc()
    assert pool.ref[src] == 2 and pool.ref[kids[0].table[-1]] == 1 and kids[0].table[0] == parent.table[0]
    for s in [parent] + kids:
        pool.release(s.table)
    assert not pool.ref and sorted(pool.free_gpu) == list(range(8))


def test_swap_moves_a_shared_block_once_and_restores_the_reference_counts():
    mod = _harness()
    pool = mod.BlockPool(8, 6)
    a = mod.Sequence(list(range(40)), 5)
    a.table = [pool.alloc(), pool.alloc(), pool.alloc()]
    b = a.fork()
    pool.share(b.table)
    b.table[-1] = pool.alloc()          # b has a last block of its own (after a copy-on-write)
    pool.ref[a.table[-1]] -= 1
    other = mod.Sequence([1, 2, 3], 2)  # an unrelated sequence keeps its block on the GPU
    other.table = [pool.alloc()]
    before = {id(s): list(s.table) for s in (a, b)}
    out = pool.swap_out([a, b])
    assert len(out) == 4 and len(set(out.values())) == 4          # 2 shared + 2 private blocks, each moved once
    assert a.table[:2] == b.table[:2] and a.table[2] != b.table[2]
    assert all(0 <= c < 6 for c in a.table + b.table)
    assert sorted(pool.ref) == other.table and len(pool.free_gpu) == 7 and len(pool.free_cpu) == 2
    assert pool.cpu_ref[a.table[0]] == 2 and pool.cpu_ref[a.table[2]] == 1
    back = pool.swap_in([a, b])
    assert set(back) == set(out.values()) and len(set(back.values())) == 4
    assert a.table[:2] == b.table[:2] and a.table[2] != b.table[2]
    assert pool.ref[a.table[0]] == 2 and pool.ref[a.table[2]] == 1 and pool.ref[b.table[2]] == 1
    assert not pool.cpu_ref and sorted(pool.free_cpu) == list(range(6))
    assert {len(v) for v in before.values()} == {3}
    for s in (a, b, other):
        pool.release(s.table)
    assert not pool.ref and sorted(pool.free_gpu) == list(range(8))
