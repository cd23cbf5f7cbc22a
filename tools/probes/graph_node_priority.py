#!/usr/bin/env python
"""Probe (review item 2c, round 5): hipGraph KERNEL-NODE PRIORITIES on the captured train step.
The step is captured with keep_graph=True; the kernel nodes are listed through the HIP graph API (ctypes on libamdhip64), named with
hipKernelNameRefByPtr, and the nodes whose kernel name matches a pattern get hipKernelNodeAttributePriority = <value> before the graph
is instantiated.  Then the step is replayed and timed like bench.py does (settle + warm-up + K steps).

    python tools/probes/graph_node_priority.py <mode> [steps]
        mode: none | low_deferred (the deferred weight gradients at the lowest priority) | high_chain (the thin critical chain at the highest)
              | both"""
import ctypes, os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from strajnet_amd import STrajNet, OGMFlow_loss, OccupancyFlowTaskConfig, Nadam
from strajnet_amd.graph import GraphedTrainStep

mode = sys.argv[1] if len(sys.argv) > 1 else 'none'
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
hip = ctypes.CDLL('libamdhip64.so')


class Dim3(ctypes.Structure):
    _fields_ = [('x', ctypes.c_uint), ('y', ctypes.c_uint), ('z', ctypes.c_uint)]


class KernelNodeParams(ctypes.Structure):
    _fields_ = [('blockDim', Dim3), ('extra', ctypes.c_void_p), ('func', ctypes.c_void_p), ('gridDim', Dim3), ('kernelParams', ctypes.c_void_p),
                ('sharedMemBytes', ctypes.c_uint)]


hip.hipKernelNameRefByPtr.restype = ctypes.c_char_p
hip.hipKernelNameRefByPtr.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
DEFERRED = re.compile(r'upconv_wgrad_tr|wgrad_sk_kernel|upconv_fold|outconv_bwd_reduce')
CHAIN = re.compile(r'xattn_bwd|fgattn_bwd|fgoff_bwd|agent_|swin_(attn|mlp)_bwd_kernel<\w+, 384|swin_split_bwd|gemm_group|linear_rs|ln_bwd')

dev = torch.device('cuda', 0)
model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0)
loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), replica=1.0, use_focal_loss=False, use_gt=True)
opt = Nadam.for_model(model, lr=1e-4)
x = bench.synth_batch(8, 1234, dev)
g = GraphedTrainStep(model, loss_fn, x, keep_graph=(mode != 'none'))
if mode != 'none':
    lo, hi = ctypes.c_int(), ctypes.c_int()
    assert hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) == 0
    print(f'stream priority range: least {lo.value} greatest {hi.value}')
    graph = ctypes.c_void_p(g.graph.raw_cuda_graph())
    n = ctypes.c_size_t()
    assert hip.hipGraphGetNodes(graph, None, ctypes.byref(n)) == 0
    nodes = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(graph, nodes, ctypes.byref(n)) == 0
    nk = nset = 0
    fails = 0
    # what the runtime accepts: read the attribute of the first kernel node, then try every value in [-3, 3] on it
    first = next(nd for nd in nodes if True)
    val = (ctypes.c_char * 64)()
    rc = hip.hipGraphKernelNodeGetAttribute(ctypes.c_void_p(first), 8, val)
    print(f'hipGraphKernelNodeGetAttribute(priority) -> rc {rc}, value {ctypes.cast(val, ctypes.POINTER(ctypes.c_int))[0]}')
    hip.hipGetLastError()
    for v in range(-3, 4):
        ctypes.cast(val, ctypes.POINTER(ctypes.c_int))[0] = v
        rc = hip.hipGraphKernelNodeSetAttribute(ctypes.c_void_p(first), 8, val)
        print(f'  set priority {v}: rc {rc}')
        hip.hipGetLastError()
    for a in (1, 2):
        rc = hip.hipGraphKernelNodeGetAttribute(ctypes.c_void_p(first), a, val)
        print(f'hipGraphKernelNodeGetAttribute(attr {a}) -> rc {rc}')
        hip.hipGetLastError()
    for node in nodes:
        t = ctypes.c_int()
        assert hip.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(t)) == 0
        if t.value != 0:
            continue
        nk += 1
        kp = KernelNodeParams()
        if hip.hipGraphKernelNodeGetParams(ctypes.c_void_p(node), ctypes.byref(kp)) != 0:
            continue
        nm = hip.hipKernelNameRefByPtr(kp.func, None)
        name = nm.decode() if nm else ''
        prio = None
        if mode in ('low_deferred', 'both') and DEFERRED.search(name):
            prio = lo.value
        if mode in ('high_chain', 'both') and CHAIN.search(name):
            prio = hi.value
        if prio is None:
            continue
        val = (ctypes.c_char * 64)()
        ctypes.cast(val, ctypes.POINTER(ctypes.c_int))[0] = prio
        rc = hip.hipGraphKernelNodeSetAttribute(ctypes.c_void_p(node), 8, val)
        if rc != 0:
            if fails == 0:
                hip.hipGetErrorString.restype = ctypes.c_char_p
                print(f'hipGraphKernelNodeSetAttribute(priority = {prio}) on {name[:60]}: error {rc} ({hip.hipGetErrorString(rc).decode()})')
            hip.hipGetLastError()          # (not sticky: the next launch check must not see it)
            fails += 1
        else:
            nset += 1
    print(f'{n.value} nodes, {nk} kernel nodes, priority set on {nset}, {fails} calls failed')
    g.graph.instantiate()


def run(k):
    for _ in range(k):
        g()
        opt.step()


run(50)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(K)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f'mode {mode}: {8 / dt:.1f} scenes/s  {dt * 1e3:.3f} ms per step', flush=True)
