"""Pick HIP streams that really run concurrently.

ROCm multiplexes HIP streams onto a small number of hardware queues (4 by default); two streams that share a queue never
overlap, and which streams share one is decided inside the runtime.  A host that keeps several independent batches in
flight (one native context per batch, SAGEN_ONE_STREAM=1 so each context stays on its caller's stream)
wants streams on distinct queues - so probe for them."""
import torch


def pick_concurrent_streams(n, candidates=12):
    """ROCm multiplexes HIP streams onto a few hardware queues; two streams on one queue never overlap.  Probe candidate
    streams pairwise (a long matmul on one, a short fill on the other: does the fill finish first?) and return n streams
    that demonstrably run concurrently with each other."""
    cand = [torch.cuda.Stream() for _ in range(candidates)]
    if n <= 1:
        return cand[:1]
    a = torch.randn(4096, 4096, device='cuda'); small = torch.empty(1024, device='cuda')
    torch.mm(a, a); torch.cuda.synchronize()

    def concurrent(s1, s2):
        e_long, e_short = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(s1)
        with torch.cuda.stream(s1):
            for _ in range(3):
                torch.mm(a, a)
        e_long.record(s1)
        with torch.cuda.stream(s2):
            small.fill_(1.0)
        e_short.record(s2)
        torch.cuda.synchronize()
        return e0.elapsed_time(e_short) < 0.5 * e0.elapsed_time(e_long)

    chosen = [cand[0]]
    for c in cand[1:]:
        if len(chosen) == n:
            break
        if all(concurrent(x, c) and concurrent(c, x) for x in chosen):
            chosen.append(c)
    if len(chosen) < n:
        chosen += [c for c in cand if c not in chosen][:n - len(chosen)]
    return chosen
