"""Latency of a cross-stream dependency inside a replayed hipGraph: a chain of N dependent ~20 us kernels, all on one stream vs
alternating between two streams (every link is then a cross-queue edge), plus the same with a THIRD busy stream next to it."""
import torch, time
dev = torch.device('cuda', 0)
a = torch.zeros(1 << 16, device=dev)
def k(): torch.cuda._sleep(40000)      # ~ 20 us
def chain(n, alt, busy):
    main = torch.cuda.current_stream()
    side = S[0]; third = S[1]
    if busy:
        third.wait_stream(main)
        with torch.cuda.stream(third):
            for _ in range(n * 4): a.add_(1.0)          # a queue that always has a short kernel ready
    cur = main
    for i in range(n):
        nxt = side if (alt and i % 2 == 1) else main
        if nxt is not cur:
            nxt.wait_stream(cur)
        with torch.cuda.stream(nxt):
            k()
        cur = nxt
    if cur is not main:
        main.wait_stream(cur)
    if busy:
        main.wait_stream(third)
S = [torch.cuda.Stream(), torch.cuda.Stream()]
cap = torch.cuda.Stream()
for busy in (False, True):
    for alt in (False, True):
        with torch.cuda.stream(cap):
            chain(40, alt, busy); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap):
                chain(40, alt, busy)
            for _ in range(3): g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10 * 1e6
            t0 = time.perf_counter()
            for _ in range(10): chain(40, alt, busy)
            torch.cuda.synchronize()
            de = (time.perf_counter() - t0) / 10 * 1e6
        print(f'busy third stream {busy!s:5}  alternating {alt!s:5}: graph {dt:7.0f} us  eager {de:7.0f} us   (40 kernels)')
