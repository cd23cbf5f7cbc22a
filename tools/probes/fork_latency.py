"""Start latency of the second branch of a fork inside a replayed hipGraph: K0 -> {chain A (nA kernels), chain B (nB kernels)} -> join.
Each kernel ~20 us.  Ideal: (1 + max(nA, nB)) * 20 us.  Variants: which chain is captured first; an extra busy branch."""
import torch, time
dev = torch.device('cuda', 0)
a = torch.zeros(1 << 16, device=dev)
def k(): torch.cuda._sleep(40000)
S = [torch.cuda.Stream() for _ in range(3)]
def body(nA, nB, first, busy):
    main = torch.cuda.current_stream()
    k()
    ev = torch.cuda.Event(); ev.record(main)
    def A():
        for _ in range(nA): k()
    def B():
        S[0].wait_event(ev)
        with torch.cuda.stream(S[0]):
            for _ in range(nB): k()
    def C():
        S[1].wait_event(ev)
        with torch.cuda.stream(S[1]):
            for _ in range(8): torch.cuda._sleep(400000)     # long kernels (like the deferred weight gradients)
    if busy: C()
    if first == 'A': A(); B()
    else: B(); A()
    main.wait_stream(S[0])
    if busy: main.wait_stream(S[1])
cap = torch.cuda.Stream()
for busy in (False, True):
    for nA, nB in ((30, 30), (30, 10), (10, 30)):
        for first in ('A', 'B'):
            with torch.cuda.stream(cap):
                body(nA, nB, first, busy); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cap):
                    body(nA, nB, first, busy)
                for _ in range(3): g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10): g.replay()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 10 * 1e6
            ideal = (1 + max(nA, nB)) * 17.6
            print(f'busy {busy!s:5} A(main)={nA:2d} B(side)={nB:2d} captured first: {first}:  graph {dt:7.0f} us   (one chain alone ~{ideal:.0f}, serialized ~{(1 + nA + nB) * 17.6:.0f})')
