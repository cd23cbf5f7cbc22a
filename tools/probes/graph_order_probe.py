"""Does a replayed hipGraph start a main-stream kernel that was CAPTURED after a side-stream chain only when that chain is done?
main: K0 ; side (forked after K0): S1 S2 S3 ; main: M1 M2 (depend on K0 only, captured AFTER S1..S3) ; join."""
import torch, time
dev = torch.device('cuda', 0)
N = 1 << 22
def busy(t, n=40):          # ~ a few tens of us on a small tensor: dependent chain of tiny kernels is not needed, one long kernel
    torch.cuda._sleep(200000)   # ~ 100 us
main_s = torch.cuda.Stream(); side = torch.cuda.Stream()
def body(order):
    main = torch.cuda.current_stream()
    torch.cuda._sleep(200000)                    # K0
    ev = torch.cuda.Event(); ev.record(main)
    def side_chain():
        side.wait_event(ev)
        with torch.cuda.stream(side):
            for _ in range(6): torch.cuda._sleep(200000)
    def main_chain():
        for _ in range(6): torch.cuda._sleep(200000)
    if order == 'side_first':
        side_chain(); main_chain()
    else:
        main_chain(); side_chain()
    main.wait_stream(side)
for order in ('side_first', 'main_first'):
    with torch.cuda.stream(main_s):
        body(order); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main_s):
            body(order)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e6
        # eager for reference
        t0 = time.perf_counter()
        for _ in range(20): body(order)
        torch.cuda.synchronize()
        de = (time.perf_counter() - t0) / 20 * 1e6
    print(order, 'graph replay %.0f us, eager %.0f us  (1 + 6 serial kernels = 7 units if the chains overlap, 13 if serialized)' % (dt, de))
