#!/usr/bin/env python
"""bench.py -- scenes/s of the STrajNet train step (fwd + OGMFlow loss + bwd [+ RCCL grad all-reduce]) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): batch 8 per GPU, bf16 storage / f32 accumulate, cfg-256
(input 256x256x11, window 8, embed 96, depths [2,2,2], heads [3,6,12], fg_msa=True, fg=True, large_ogm=False),
8 waypoints, observed + occluded + flow heads, synthetic inputs of SURVEY.md 8(d), random-init weights.
A "step" = model forward, OGMFlow_loss (use_gt AUC gate + flow-warp term), backward into the flat f32 gradient
buffer, and -- for N>1 -- one RCCL SUM all-reduce of that buffer (loss pre-scaled by 1/N via replica=N, exactly as
loss.py:200 + MirroredStrategy do).  The optimizer is not part of the metric ("next" item 8f-1).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_GFLOP_FWD_PER_SCENE = 200.7      # SURVEY.md App. E (direct form, 2/MAC)
ALGO_GFLOP_STEP_PER_SCENE = 602.0     # fwd + dgrad + wgrad
ALGO_GFLOP_STEP_PER_SCENE_512 = 3 * 240.9   # cfg-512 with depths [2,2,6]
PEAK_BF16_TFLOPS = 2500.0             # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3

CFG256 = dict(input_size=(256, 256), window_size=8, embed_dim=96, depths=[2, 2, 2], num_heads=[3, 6, 12])


def synth_batch(B, seed, device, grid=256):
    """Synthetic scene batch, generated with torch on the host then moved to HBM.  grid = 512: the cfg-512
    geometry (512^2 ogm / flow rasters, 256^2 map image, 256^2 ground truth)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    H = 256
    ogm = (torch.rand((B, grid, grid, 11, 2), generator=g) < 0.02).float()
    flow = torch.randn((B, grid, grid, 2), generator=g) * 2.0 * ogm[..., 10, 0:1]
    map_img = torch.randint(-128, 128, (B, H, H, 3), generator=g).float() / 256.0

    def agents(n):
        a = torch.zeros((B, n, 11, 8))
        a[..., 0:2] = torch.rand((B, n, 11, 2), generator=g) * 80 - 40
        a[..., 2:4] = torch.randn((B, n, 11, 2), generator=g) * 5
        a[..., 4] = torch.rand((B, n, 11), generator=g) * 6.283 - 3.1416
        ty = torch.randint(0, 3, (B, n), generator=g)
        for k in range(3):
            a[..., 5 + k] = (ty == k).float()[..., None]
        a[:, n - n // 4:] = 0
        return a
    obs, occ = agents(48), agents(16)
    gt_obs = (torch.rand((B, 8, H, H, 1), generator=g) < 0.02).float()
    gt_occ = (torch.rand((B, 8, H, H, 1), generator=g) < 0.005).float()
    either = torch.maximum(gt_obs, gt_occ)
    gt_flow = torch.randn((B, 8, H, H, 2), generator=g) * 3 * either
    origin = torch.rand((B, 8, H, H, 1), generator=g) * (torch.rand((B, 8, H, H, 1), generator=g) < 0.03).float()
    origin = torch.maximum(origin, 0.9 * either * (torch.rand((B, 8, H, H, 1), generator=g) < 0.5).float())
    d = dict(ogm=ogm, flow=flow, map_img=map_img, obs=obs, occ=occ, mapt=torch.zeros((B, 256, 10, 7)),
             gt_obs=gt_obs, gt_occ=gt_occ, gt_flow=gt_flow, origin_flow=origin)
    return {k: v.to(device) for k, v in d.items()}


def bench_infer(args, model, x, world, rank, dist):
    """BASELINE config 4: replicas only (no collective), one hipGraph replay of the eval forward per step."""
    import torch
    from strajnet_amd.graph import GraphedForward
    gf = None if args.no_graph else GraphedForward(model, x)

    def step():
        if gf is not None:
            return gf()
        with torch.no_grad():
            return model(x['ogm'], x['map_img'], training=False, obs=x['obs'], occ=x['occ'], mapt=None, flow=x['flow'])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt_s = time.perf_counter() - t0
    t = torch.tensor([dt_s], device=out.device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_s = float(t)
    if rank == 0:
        B = args.batch
        print(json.dumps({
            'metric': 'scenes/sec (inference forward, 256x256 grids)', 'value': round(B * world * args.steps / dt_s, 3), 'unit': 'scenes/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt_s / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'STrajNet {"cfg-512" if args.cfg512 else "cfg-256"} inference forward (BASELINE config 4: extra measurement, not the headline metric), '
                                   f'batch {B}/GPU, fg_msa+fg, random-init weights', 'global_batch': B * world, 'parallelism': f'replicas x{world}',
                       'hipgraph': gf is not None, 'finite': bool(torch.isfinite(out).all())}}))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(max_seconds=30.0):
    """The oracle's PyTorch-CPU restatement ("port", NOT the TensorFlow reference) timed on this box's host cores:
    B=1 cfg-256 f32 forward + loss + backward, 1 warm-up, then timed steps until ~12 s of CPU work (bounded by max_seconds)."""
    import numpy as np
    import torch
    from oracle import np_ref, torch_ref
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))          # more threads than this only oversubscribes the small ops of the port
    torch.set_num_threads(cores)
    w = np_ref.make_weights(CFG256, 0, mode='reference')
    x = np_ref.make_inputs(CFG256, 1)
    p = torch_ref.to_torch(w, torch.float32, requires_grad=True)
    xt = torch_ref.to_torch(x, torch.float32)

    def step():
        for v in p.values():
            v.grad = None
        y = torch_ref.forward(p, CFG256, xt['ogm'], xt['map_img'], xt['obs'], xt['occ'], xt['flow'])
        d = torch_ref.loss(y, xt['gt_obs'], xt['gt_occ'], xt['gt_flow'], xt['origin_flow'])
        sum(d.values()).backward()
    t0 = time.time()
    step()
    warm = time.time() - t0
    times = []
    while (sum(times) < 12.0 or len(times) < 2) and len(times) < 64 and (time.time() - t0) < max_seconds:
        t1 = time.time()
        step()
        times.append(time.time() - t1)
    if not times:
        times = [warm]
    sec = float(np.median(times))
    return {'value': round(1.0 / sec, 4), 'unit': 'scenes/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'B=1 cfg-256 f32 fwd+loss+bwd, oracle/torch_ref.py on CPU (not TensorFlow), 1 warm-up + {len(times)} timed steps '
                      f'({sum(times):.1f} s of CPU work), median {sec:.2f} s/step'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8, help='scenes per GPU')
    ap.add_argument('--dtype', default=None, choices=['bf16', 'f32', 'f16'], help='default bf16 (train step) / f16 (--infer)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true', help='skip every extra pass after the timed region (per-kernel HIP-event timing, the optimizer-free repeat): the run then executes exactly warmup + steps steps (+ 2 capture warm-ups), which is what the rocprofv3 summaries divide by')
    ap.add_argument('--no-graph', action='store_true', help='run the step eagerly instead of replaying the captured hipGraph')
    ap.add_argument('--cfg512', action='store_true', help='BASELINE config 5 instead of the metric config: 512x512 rasters, large_ogm, depths [2,2,6] (extra measurement, not the headline)')
    ap.add_argument('--infer', action='store_true', help='BASELINE config 4 instead of the metric config: inference-only forward, batch 32/GPU, fp16 MFMA path, hipGraph replay (extra measurement, not the headline)')
    ap.add_argument('--no-optimizer', action='store_true', help='time fwd + loss + bwd (+ all-reduce) only, without the fused Keras-Nadam update (train.py:197,224) that the default step ends with')
    ap.add_argument('--serial', action='store_true', help='no side streams: every kernel runs alone (the mode the roofline kernel timings are taken in)')
    ap.add_argument('--gemm-trace', action='store_true', help='print per-shape GEMM launch times (HIP events) to stderr')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from strajnet_amd import STrajNet, OGMFlow_loss, OccupancyFlowTaskConfig, get_pred_waypoint_logits, warpped_gt
    from strajnet_amd import ops

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world == 1:
        print('bench.py: --gpus N>1 must be launched through torch.distributed.run', file=sys.stderr)
        sys.exit(2)
    # STJ_BENCH_SHARE_GPU=1 (test hook): all ranks on cuda:0 with the gloo backend -- exercises the multi-process code path of
    # this file on a 1-GPU box (RCCL refuses two ranks on one device); never set for real measurements
    share = os.environ.get('STJ_BENCH_SHARE_GPU') == '1'
    dev = torch.device('cuda', 0 if share else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
        # communicator set-up (seconds the first time) must never land in the timed region, whatever --warmup says
        dist.all_reduce(torch.zeros(1, device=dev))
        torch.cuda.synchronize()

    if args.dtype is None:
        args.dtype = 'f16' if args.infer else 'bf16'
    if args.dtype == 'f16' and not args.infer:
        print('bench.py: fp16 is the inference mode (no loss scaling); use --infer', file=sys.stderr)
        sys.exit(2)
    dtype = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f16': torch.float16}[args.dtype]
    cfg = dict(CFG256, input_size=(512, 512), depths=[2, 2, 6]) if args.cfg512 else CFG256
    model = STrajNet(cfg, fg_msa=True, fg=True, large_ogm=args.cfg512, dtype=dtype, device=dev, seed=0)
    loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), ogm_weight=1000.0, occ_weight=1000.0, flow_weight=1.0,
                           replica=float(world), flow_origin_weight=1000.0, no_use_warp=False, use_pred=False,
                           use_focal_loss=False, use_gt=True)
    model.serial = args.serial
    if args.infer and args.batch == 8:
        args.batch = 32
    B = args.batch
    x = synth_batch(B, 1234 + rank, dev, 512 if args.cfg512 else 256)
    if args.infer:
        return bench_infer(args, model, x, world, rank, dist)

    from strajnet_amd import Nadam
    opt = None if args.no_optimizer else Nadam.for_model(model, lr=1e-4)        # train.py:197

    def finish_step():
        # the exchange step of data parallelism (one SUM all-reduce of the flat f32 gradient bucket), then the optimizer on it
        if world > 1:
            dist.all_reduce(model.flat_grads(), op=dist.ReduceOp.SUM)
        if opt is not None:
            opt.step()

    def step():
        model.zero_grad()
        out = model(x['ogm'], x['map_img'], training=True, obs=x['obs'], occ=x['occ'], mapt=x['mapt'], flow=x['flow'])
        d = loss_fn(get_pred_waypoint_logits(out), warpped_gt(x['gt_obs'], x['gt_occ'], x['gt_flow'], x['origin_flow']), None)
        total = d['observed_xe'] + d['occluded_xe'] + d['flow'] + d['flow_warp_xe']
        total.backward()
        finish_step()
        return total

    graphed = None
    if not args.no_graph:
        try:
            from strajnet_amd.graph import GraphedTrainStep
            graphed = GraphedTrainStep(model, loss_fn, x)
        except Exception as e:           # capture is an optimisation, never a requirement
            print(f'bench.py: hipGraph capture failed ({type(e).__name__}: {e}); running eagerly', file=sys.stderr)
            graphed = None
    eager_step = step

    def step_graph():
        losses = graphed()
        finish_step()
        return losses.sum()
    if graphed is not None:
        step = step_graph

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    barrier()
    dt_s = time.perf_counter() - t0
    # SURVEY 8(d) words the metric as fwd + loss + bwd (+ all-reduce) WITHOUT the optimizer: the same K steps again with the
    # Nadam launch left out, reported next to the headline (which includes it)
    fwd_bwd_only = None
    if opt is not None and not args.no_kernel_timing:
        held, opt = opt, None
        step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        d1 = time.perf_counter() - t1
        opt = held
        if world > 1:
            t = torch.tensor([d1], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d1 = float(t.item())
        fwd_bwd_only = {'value': round(B * world * args.steps / d1, 3), 'unit': 'scenes/s', 'ms_per_step': round(d1 / args.steps * 1e3, 3)}
    # the exchange step on its own (measured all-reduce time next to the ring model), outside the timed region
    allreduce = None
    if world > 1:
        g = model.flat_grads()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        barrier()
        e0.record()
        for _ in range(5):
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        nbytes = g.numel() * 4
        allreduce = {'ms': round(ms, 4), 'mbytes': round(nbytes / 1e6, 1), 'collective': 'all_reduce(sum, f32), one bucket after backward',
                     'busbw_GBps': round(2 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9, 1),
                     'frac_of_step': round(ms / (dt_s / args.steps * 1e3), 4)}
    # per-kernel HIP-event timing of the conv kernels (roofline of the dominant one): the same launches, issued eagerly
    # with events recorded on the launch stream -- under graph replay individual launches cannot carry events
    prof = prof_conc = None
    if not args.no_kernel_timing:
        # (a) kernels ALONE: side streams off, so an event pair brackets exactly one kernel -- the roofline figures
        model.serial = True
        ops.PROF_GEMM = args.gemm_trace
        eager_step()
        ops.prof_enable()
        for _ in range(min(args.steps, 3)):
            eager_step()
        barrier()
        prof = ops.prof_disable()
        # (b) as scheduled in the measured step (branches on concurrent streams share the GPU): reported next to (a)
        model.serial = args.serial
        ops.PROF_GEMM = False
        eager_step()
        ops.prof_enable()
        for _ in range(min(args.steps, 3)):
            eager_step()
        barrier()
        prof_conc = ops.prof_disable()
    if world > 1:
        t = torch.tensor([dt_s], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_s = float(t.item())
    loss_val = float(last.detach())

    if rank == 0:
        scenes = B * world * args.steps
        algo_step = ALGO_GFLOP_STEP_PER_SCENE_512 if args.cfg512 else ALGO_GFLOP_STEP_PER_SCENE
        value = scenes / dt_s
        peak = PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_TFLOPS
        roof = None
        kern = {}
        if prof:
            for k, evs in prof.items():
                ms = [a.elapsed_time(b) for a, b, _ in evs]
                kern[k] = (sum(ms) / len(ms), evs[0][2], sum(ms))
            if args.gemm_trace:
                tot = sum(v[2] for k, v in kern.items() if k.startswith('gemm')) / min(args.steps, 3)
                print(f'# GEMM launches by shape: {tot:.3f} ms/step', file=sys.stderr)
                for k in sorted((k for k in kern if k.startswith('gemm')), key=lambda k: -kern[k][2]):
                    n = len(prof[k]) / min(args.steps, 3)
                    print(f'{kern[k][2] / min(args.steps, 3) * 1e3:9.1f} us/step  {n:5.1f} x {kern[k][0] * 1e3:7.1f} us  '
                          f'{kern[k][1] / kern[k][0] / 1e9:7.1f} TF/s  {k}', file=sys.stderr)
                kern = {k: v for k, v in kern.items() if not k.startswith('gemm')}
            dom = max(kern, key=lambda k: kern[k][2])
            avg_ms, flops, _ = kern[dom]
            ach = flops / (avg_ms * 1e-3) / 1e12
            traffic = None
            tpath = os.path.join(ROOT, 'profiles', 'roofline_traffic.json')
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(dom)
                except Exception:
                    traffic = None
            conc = prof_conc.get(dom) if prof_conc else None
            conc_ms = sum(a.elapsed_time(b) for a, b, _ in conc) / len(conc) if conc else None
            roof = {'bound': 'mfma', 'kernel': dom, 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': round(ach / peak, 4), 'traffic': traffic, 'avg_launch_ms': round(avg_ms, 4),
                    'timing': 'kernel alone on the GPU (side streams off, eager pass after the timed region)',
                    'avg_launch_ms_in_step': round(conc_ms, 4) if conc_ms else None,
                    'algorithmic_gflop_per_launch': round(flops / 1e9, 2),
                    'end_to_end_frac': round(value * algo_step / 1e3 / (peak * world), 4)}
        out = {
            'metric': 'scenes/sec (fwd+bwd, 256x256 grids)', 'value': round(value, 3), 'unit': 'scenes/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt_s / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'STrajNet {"cfg-512 (large_ogm, depths [2,2,6])" if args.cfg512 else "cfg-256"} train step (fwd+OGMFlow_loss+bwd{"+RCCL grad all-reduce" if world > 1 else ""}{"+Nadam" if opt is not None else ""}), '
                                   f'batch {B}/GPU, 8 waypoints, obs+occ+flow heads, fg_msa+fg, random-init weights',
                       'global_batch': B * world, 'grid': '256x256x11', 'parallelism': f'dp{world}', 'hipgraph': graphed is not None, 'concurrent_branch_streams': not args.serial,
                       'optimizer_in_step': opt is not None, 'algorithmic_gflop_per_scene_step': algo_step},
            'loss': round(loss_val, 4),
            'without_optimizer': fwd_bwd_only,
            'allreduce': allreduce,
            'roofline': roof,
        }
        if prof:
            out['kernel_ms'] = {k: round(v[0], 4) for k, v in sorted(kern.items())}
        if not args.no_cpu_baseline and world == 1 and not args.cfg512:
            try:
                out['cpu_baseline'] = cpu_baseline()
            except Exception as e:      # the CPU port is a reported extra, never the product path
                out['cpu_baseline'] = {'value': None, 'unit': 'scenes/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': f'failed: {e}'}
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
