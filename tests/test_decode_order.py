"""The balanced work order of the decode kernels, restated in Python and
checked for what the kernels rely on: with the row table that the plan kernel
builds, every (sequence, head) row is visited by exactly one workgroup of the
grid the host launches, for any batch size, head count and length mixture, and
the XCDs' shares come out even with each XCD's longest rows first.  (The
kernels themselves are checked on the GPU in test_decode_gpu.py: every output
row written and equal to the oracle's.)"""
import numpy as np
import pytest


def plan(S, H, balance_unit=4):
    group = max(1, min(16, 128 // H))
    unit = min(group, balance_unit)
    group = group // unit * unit
    units = (S + unit - 1) // unit
    slots = ((units + 7) // 8 * unit + group - 1) // group * group
    return group, unit, units, slots


def row_table(blocks, unit, slots):
    """pa_decode_plan_kernel, step 3: [8][slots] sequence numbers, -1 = empty."""
    S = len(blocks)
    n_units = (S + unit - 1) // unit
    seq_key = [(min(b, (1 << 20) - 1) << 10) | (1023 - s) for s, b in enumerate(blocks)]
    unit_key = [(min(sum(blocks[u * unit:(u + 1) * unit]), (1 << 22) - 1) << 10) | (1023 - u) for u in range(n_units)]
    full_rows = n_units >> 3
    seq_xcd = [None] * S
    tail = []
    for u in range(n_units):
        rank = sum(k > unit_key[u] for k in unit_key)
        row, col = divmod(rank, 8)
        for s in range(u * unit, min((u + 1) * unit, S)):
            if row < full_rows:
                seq_xcd[s] = 7 - col if row & 1 else col
            else:
                tail.append(s)  # units of the partial last row: their sequences are dealt one by one
    for s in tail:
        rank = sum(seq_key[t] > seq_key[s] for t in tail)
        row, col = full_rows + rank // 8, rank % 8
        seq_xcd[s] = 7 - col if row & 1 else col
    table = [[-1] * slots for _ in range(8)]
    for s in range(S):
        x = seq_xcd[s]
        place = sum(seq_key[t] > seq_key[s] for t in range(S) if seq_xcd[t] == x)
        assert table[x][place] == -1
        table[x][place] = s
    return table


def rows_visited(S, H, blocks):
    """(seq, head) of every row workgroup that does not exit, by XCD, in launch order."""
    G, U, units, NX = plan(S, H)
    assert units >= 16 and S <= 1024 and 8 * NX <= 8 * (1024 // 8 + 32)
    table = row_table(blocks, U, NX)
    out = {x: [] for x in range(8)}
    for b in range(8 * NX * H):
        x, w = b & 7, b >> 3
        c, inn = divmod(w, G * H)
        hd, j = divmod(inn, G)
        seq = table[x][c * G + j]
        if seq >= 0:
            out[x].append((seq, hd))
    return out


@pytest.mark.parametrize("H", [1, 8, 12, 32, 40, 64, 128])
def test_balanced_order_visits_every_row_once(H):
    rng = np.random.default_rng(H)
    _, U, _, _ = plan(64, H)
    for S in sorted({16 * U - U + 1, 16 * U, 16 * U + 1, 24 * U + 2, 255, 256, 257, 1000, 1024}):
        if plan(S, H)[2] < 16:
            continue
        blocks = rng.integers(0, 130, size=S).tolist()
        seen = [row for rows in rows_visited(S, H, blocks).values() for row in rows]
        assert len(seen) == S * H and len(set(seen)) == S * H, (S, H)


def test_balanced_order_evens_out_the_xcd_shares_and_runs_longest_first():
    rng = np.random.default_rng(1)
    S, H = 256, 32
    ratios = []
    for _ in range(20):
        blocks = np.minimum(rng.lognormal(np.log(20), 1.0, size=S).astype(int) + 1, 128).tolist()
        visited = rows_visited(S, H, blocks)
        per_xcd = [sum(blocks[s] for s, _ in rows) for rows in visited.values()]
        ratios.append(max(per_xcd) / (sum(per_xcd) / 8))
        for rows in visited.values():
            order = list(dict.fromkeys(s for s, _ in rows))  # sequences in first-visit order
            chunks = [order[i:i + 4] for i in range(0, len(order), 4)]
            assert all(min(blocks[s] for s in a) >= max(blocks[s] for s in b) for a, b in zip(chunks, chunks[1:]))
    # batch order deals 1.26 on average (up to 1.5) for these lengths
    assert np.mean(ratios) < 1.08 and max(ratios) < 1.2


@pytest.mark.parametrize("S,H", [(100, 40), (100, 32), (200, 32), (77, 12), (61, 32)])
def test_balanced_order_of_equal_lengths_gives_no_xcd_more_than_its_share(S, H):
    """Equal lengths: the fullest XCD holds ceil(S / 8) sequences, not a whole unit more."""
    counts = [len(rows) // H for rows in rows_visited(S, H, [64] * S).values()]
    assert sum(counts) == S and max(counts) == -(-S // 8), counts
