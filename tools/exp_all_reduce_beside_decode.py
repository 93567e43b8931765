#!/usr/bin/env python3
"""What the one-shot all-reduce costs a decode kernel that runs beside it on the same GPU: two
ranks of a communicator on two side streams of one process reduce a [256, 4096] fp16 buffer over
and over while the main stream runs decode launches (256 x 32 x 128, context 1024).

    python tools/exp_all_reduce_beside_decode.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__  # noqa: E402


def main():
    pkg = __graft_entry__.load_package()
    from vllm_amd.worker import PushAllReduce
    c = pkg._capi
    dev = torch.device("cuda:0")
    side = [torch.cuda.Stream(dev) for _ in range(2)]   # created first: hardware queues of their own
    S, H, D, B, CTX, L = 256, 32, 128, 16, 1024, 4
    bps = CTX // B
    NB = S * bps
    g = torch.Generator(device="cpu").manual_seed(0)
    caches = [(torch.empty(NB, H, D // 8, B, 8, dtype=torch.float16, device=dev).normal_(0, 0.5),
               torch.empty(NB, H, D, B, dtype=torch.float16, device=dev).normal_(0, 0.5)) for _ in range(L)]
    bt = torch.randperm(NB, generator=g).to(torch.int32).reshape(S, bps).to(dev)
    cl = torch.full((S,), CTX, dtype=torch.int32, device=dev)
    q = torch.empty(S, H, D, dtype=torch.float16, device=dev).normal_(0, 0.5)
    out = torch.empty_like(q)
    ws = torch.empty(max(c.decode_workspace_bytes(S, H, D, B, CTX), 8), dtype=torch.uint8, device=dev)
    c.single_query_cached_kv_attention_planned(out, q, caches[0][0], caches[0][1], D ** -0.5, bt, cl, B, CTX, ws, False, S * CTX)

    def decode(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(n):
            kc, vc = caches[i % L]
            c.single_query_cached_kv_attention_planned(out, q, kc, vc, D ** -0.5, bt, cl, B, CTX, ws, True, S * CTX)
        b.record()
        return a, b

    for algorithm, name in ((PushAllReduce.ONE_SHOT, "one-shot"), (PushAllReduce.TWO_SHOT, "two-shot")):
        ranks = PushAllReduce.local_ranks(2, S * 4096 * 2, algorithm=algorithm)
        bufs = [torch.zeros(S, 4096, dtype=torch.float16, device=dev) for _ in range(2)]
        decode(8)
        torch.cuda.synchronize()
        a, b = decode(40)
        torch.cuda.synchronize()
        alone = a.elapsed_time(b) / 40 * 1e3
        n_reduce = 400
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2)]
        for r in range(2):
            with torch.cuda.stream(side[r]):
                ev[r][0].record()
        # interleave the two ranks' launches so that neither queue runs far ahead of the other
        for _ in range(n_reduce):
            for r in range(2):
                with torch.cuda.stream(side[r]):
                    ranks[r](bufs[r])
        for r in range(2):
            with torch.cuda.stream(side[r]):
                ev[r][1].record()
        a, b = decode(40)
        torch.cuda.synchronize()
        beside = a.elapsed_time(b) / 40 * 1e3
        per_reduce = ev[0][0].elapsed_time(ev[0][1]) / n_reduce * 1e3
        done, gave_up = ranks[0].status()
        print(f"{name}: decode launch alone {alone:7.1f} us, beside the all-reduces {beside:7.1f} us; "
              f"one 2 MiB reduce between two ranks on this GPU {per_reduce:6.1f} us; reduces done {done}, waits given up {gave_up}",
              flush=True)
        for r in ranks:
            r.close()


main()
