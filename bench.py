#!/usr/bin/env python
"""bench.py -- scenes/s of the STrajNet train step (fwd + OGMFlow loss + bwd [+ RCCL grad all-reduce]) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): batch 8 per GPU, bf16 storage / f32 accumulate, cfg-256
(input 256x256x11, window 8, embed 96, depths [2,2,2], heads [3,6,12], fg_msa=True, fg=True, large_ogm=False),
8 waypoints, observed + occluded + flow heads, synthetic inputs of SURVEY.md 8(d), random-init weights.
A "step" = model forward (training=True draws), OGMFlow_loss (use_gt AUC gate + flow-warp term), backward into the flat f32
gradient buffer, for N>1 the RCCL SUM all-reduce of that buffer (loss pre-scaled by 1/N via replica=N, exactly as loss.py:200 +
MirroredStrategy do; the decoder / attention bucket is reduced under the encoder's backward), and the fused Keras-Nadam update
(train.py:197,224) -- `value` INCLUDES the optimizer (`config.optimizer_in_step`); SURVEY 8d words the metric without it, so
the optimizer-free rate of the same run is reported next to it (`without_optimizer`; `--no-optimizer` times only that).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line; besides the contract fields it carries
`roofline` (the (kernel, shape) with the largest total time in the step, timed alone with HIP events), `families` (every kernel
family: ms/step, launches, executed TFLOP/s, algorithmic GB/s), `parity_mode` / `bf16_error` (the f32 mode that holds the 1e-3
gate: its rate and error vs the CPU oracle; the timed bf16 mode's error vs that f32 mode) and `cpu_baseline`.
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
PEAK_HBM_GBPS = 8000.0                # HBM3E (MI355X_MICROARCH.md)
MFMA_LOOP_RATE_TFLOPS = 1950.0        # what a bare 16x16x32 bf16 MFMA loop sustains, 8 waves per CU (tools/probes/clock_probe.hip): informational
HBM_STREAM_CEILING_GBPS = 6750.0      # what a pure read stream reaches with every CU streaming (tools/probes/dma_lgkm_probe.hip): informational

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


def build_roofline(prof_ser, prof_conc, nprof, value, B, world, algo_gflop_per_scene, peak, gemm_trace=False):
    """-> (roofline dict of the (kernel, shape) with the largest total time per step, families table, kernel_ms table) from the
    per-launch HIP-event records of strajnet_amd/prof.py (serial pass = every kernel alone; concurrent pass = as scheduled)."""
    if not prof_ser:
        return None, None, {}
    from strajnet_amd import prof as kprof
    keys, fams = kprof.summarize(prof_ser, nprof)
    ckeys = kprof.summarize(prof_conc, nprof)[0] if prof_conc else {}
    if gemm_trace:
        print('# launches by kernel and shape (serial pass)', file=sys.stderr)
        for k in sorted(keys, key=lambda k: -keys[k]['ms_per_step']):
            v = keys[k]
            print(f"{v['ms_per_step'] * 1e3:9.1f} us/step  {v['launches']:5.1f} x {v['avg_ms'] * 1e3:7.1f} us  "
                  f"{v['flops_exec'] / v['avg_ms'] / 1e9:7.1f} TF/s  {v['bytes_alg'] / v['avg_ms'] / 1e6:7.0f} GB/s  {k}", file=sys.stderr)
    tot_ms = sum(f['ms_per_step'] for f in fams.values())
    families = {}
    for n in sorted(fams, key=lambda n: -fams[n]['ms_per_step']):
        f = fams[n]
        families[n] = {'ms_per_step': round(f['ms_per_step'], 4), 'launches': round(f['launches'], 1),
                       'TFLOPs_exec': round(f['flop_exec'] / f['ms_per_step'] / 1e9, 1) if f['flop_exec'] else None,
                       'GBps': round(f['bytes'] / f['ms_per_step'] / 1e6, 0) if f['bytes'] else None}
    # the dominant kernel = the (kernel, shape) with the largest total time in the step, whatever its family
    dom = max(keys, key=lambda k: keys[k]['ms_per_step'])
    d = keys[dom]
    t_s = d['avg_ms'] * 1e-3
    ai_exec = d['flops_exec'] / max(d['bytes_alg'], 1.0)
    balance = peak * 1e12 / (PEAK_HBM_GBPS * 1e9)
    bound = 'mfma' if ai_exec >= balance else 'hbm'
    tf_alg, tf_exec, gbps = d['flops_alg'] / t_s / 1e12, d['flops_exec'] / t_s / 1e12, d['bytes_alg'] / t_s / 1e9
    traffic = tsrc = None
    tpath = os.path.join(ROOT, 'profiles', 'roofline_traffic.json')
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(dom)
            tsrc = tj.get('_snapshot')
        except Exception:
            traffic = None
    exec_gf_step = sum(f['flop_exec'] for f in fams.values()) / 1e9
    roof = {'bound': bound, 'kernel': dom,
            # an MFMA-bound line is priced on the FLOPs the kernel EXECUTES (the phase fold removes 2.25x of the algorithmic 3x3-on-upsampled
            # count); the algorithmic rate stays next to it as mfma_frac_algorithmic
            'achieved': round(gbps if bound == 'hbm' else tf_exec, 2), 'peak': PEAK_HBM_GBPS if bound == 'hbm' else peak,
            'unit': 'GB/s' if bound == 'hbm' else 'TFLOP/s',
            'frac': round(gbps / PEAK_HBM_GBPS if bound == 'hbm' else tf_exec / peak, 4),
            'traffic': traffic,
            'traffic_source': (f'kept PMC figure (HBM bytes per launch, rocprofv3 --pmc passes of {tsrc or "an earlier snapshot"}: profiles/roofline_traffic.json), '
                               'not measured in this run') if traffic else None,
            'mfma_frac_algorithmic': round(tf_alg / peak, 4), 'mfma_frac_executed': round(tf_exec / peak, 4),
            'hbm_frac': round(gbps / PEAK_HBM_GBPS, 4),
            'hbm_frac_pmc': round(traffic / t_s / 1e9 / PEAK_HBM_GBPS, 4) if traffic else None,
            # (informational: the same two ratios against the read-stream rate measured on this part, not against the 8 TB/s peak `frac` uses)
            'hbm_frac_of_measured_stream_ceiling': round(gbps / HBM_STREAM_CEILING_GBPS, 4),
            'mfma_frac_executed_of_measured_loop_rate': round(tf_exec / MFMA_LOOP_RATE_TFLOPS, 4) if peak >= 2000 else None,
            'hbm_frac_pmc_of_measured_stream_ceiling': round(traffic / t_s / 1e9 / HBM_STREAM_CEILING_GBPS, 4) if traffic else None,
            'executed_flop_per_byte': round(ai_exec, 1), 'machine_balance_flop_per_byte': round(balance, 1),
            'avg_launch_ms': round(d['avg_ms'], 4), 'launches_per_step': round(d['launches'], 1),
            'ms_per_step_of_this_kernel': round(d['ms_per_step'], 4), 'serial_kernel_ms_per_step': round(tot_ms, 3),
            'timing': 'kernel alone on the GPU (side streams off, eager pass after the timed region, HIP events on the launch stream)',
            'workgroup_budget': ('the two large up-conv weight-gradient launches take 256 workgroups when alone (this timing) and 128 -- half '
                                 'the CUs -- in the timed step, where they are deferred next to chains of short kernels: avg_launch_ms_in_step '
                                 'is the duration there (the wg_budget argument of stj_upconv_wgrad)') if dom.startswith('upconv_wgrad[') else None,
            'avg_launch_ms_in_step': round(ckeys[dom]['avg_ms'], 4) if dom in ckeys else None,
            'algorithmic_gflop_per_launch': round(d['flops_alg'] / 1e9, 2), 'executed_gflop_per_launch': round(d['flops_exec'] / 1e9, 2),
            'algorithmic_mbytes_per_launch': round(d['bytes_alg'] / 1e6, 1),
            'end_to_end_frac': round(value * algo_gflop_per_scene / 1e3 / (peak * world), 4),
            'end_to_end_frac_executed': round(value / B * exec_gf_step / 1e3 / (peak * world), 4),
            'executed_gflop_per_scene_step': round(exec_gf_step / B, 1)}
    kern = {k: round(v['avg_ms'], 4) for k, v in sorted(keys.items()) if v['ms_per_step'] >= 0.1}
    return roof, families, kern


def bench_infer(args, model, x, world, rank, dist):
    """BASELINE config 4: replicas only (no collective), one hipGraph replay of the eval forward per step."""
    import torch
    from strajnet_amd.graph import GraphedForward
    pipe = not args.no_agent_pipeline
    gf = None if args.no_graph else GraphedForward(model, x, pipeline_agents=pipe)

    def step():
        if gf is not None:
            out = gf()
            gf.prefetch_agents()         # the NEXT batch's agent branch (its own graph, its own stream) runs under this batch's raster path
            return out
        with torch.no_grad():
            return model(x['ogm'], x['map_img'], training=False, obs=x['obs'], occ=x['occ'], mapt=None, flow=x['flow'])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(max(0, args.settle) + args.warmup):        # (settle: set-up replays, see main())
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
    single = None
    # (not under --no-kernel-timing: that run executes exactly warmup + steps replays, which is what the rocprofv3 summaries divide by and what
    #  tools/timeline.py takes its step from)
    if gf is not None and pipe and not args.no_kernel_timing:          # the same without the overlap: every step runs its own agent branch in front of the main graph
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            out = gf()
        barrier()
        single = (time.perf_counter() - t1) / args.steps * 1e3
    roof = families = None
    if not args.no_kernel_timing:        # every C-ABI launch of the forward timed alone (serial eager pass), as in the train leg
        from strajnet_amd import prof as kprof
        model.serial = True

        def eager():
            with torch.no_grad():
                return model(x['ogm'], x['map_img'], training=False, obs=x['obs'], occ=x['occ'], mapt=None, flow=x['flow'])
        eager()
        kprof.enable()
        for _ in range(3):
            eager()
        barrier()
        prof_ser = kprof.disable()
        model.serial = args.serial
        value = args.batch * world * args.steps / dt_s
        peak = PEAK_BF16_TFLOPS if args.dtype != 'f32' else PEAK_F32_TFLOPS
        algo = (240.9 if args.cfg512 else ALGO_GFLOP_FWD_PER_SCENE)
        roof, families, _ = build_roofline(prof_ser, None, 3, value, args.batch, world, algo, peak)
    if rank == 0:
        B = args.batch
        print(json.dumps({
            'metric': 'scenes/sec (inference forward, 256x256 grids)', 'value': round(B * world * args.steps / dt_s, 3), 'unit': 'scenes/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt_s / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'STrajNet {"cfg-512" if args.cfg512 else "cfg-256"} inference forward (BASELINE config 4: extra measurement, not the headline metric), '
                                   f'batch {B}/GPU, fg_msa+fg, random-init weights', 'global_batch': B * world, 'parallelism': f'replicas x{world}',
                       'hipgraph': gf is not None, 'finite': bool(torch.isfinite(out).all()),
                       'agent_pipeline': ('the agent branch of batch i + 1 (taken out of the captured forward, launched on the model\'s agent-branch stream) runs under the raster '
                                          'path of batch i' + (f'; ms_per_step with every batch running its own agent branch first: {single:.3f}' if single is not None else '')) if (gf is not None and pipe) else False},
            'roofline': roof, 'families': families}))
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
    with torch.no_grad():
        y_ref = torch_ref.forward(p, CFG256, xt['ogm'], xt['map_img'], xt['obs'], xt['occ'], xt['flow']).detach().numpy().copy()
    times = []
    while (sum(times) < 12.0 or len(times) < 2) and len(times) < 64 and (time.time() - t0) < max_seconds:
        t1 = time.time()
        step()
        times.append(time.time() - t1)
    if not times:
        times = [warm]
    sec = float(np.median(times))
    return ({'value': round(1.0 / sec, 4), 'unit': 'scenes/s', 'cores': torch.get_num_threads(), 'kind': 'port',
             'sample': f'B=1 cfg-256 f32 fwd+loss+bwd, oracle/torch_ref.py on CPU (not TensorFlow), 1 warm-up + {len(times)} timed steps '
                       f'({sum(times):.1f} s of CPU work), median {sec:.2f} s/step'}, y_ref, w, x)


def parity_extras(model, loss_fn, x, B, dev, with_cpu):
    """After the timed region, same process: (1) `parity_mode` -- the f32-storage mode (the one that holds north_star's 1e-3
    abs gate) timed on the same batch, and, inside the cpu_baseline leg (the only place bench.py touches oracle/), its max-abs
    error against the CPU oracle's forward on the oracle's own B=1 inputs; (2) `bf16_error` -- the timed bf16 mode against that
    f32 mode on the bench batch (eval forward): max-abs / rms of the logits and the largest PR-AUC difference."""
    import torch
    from strajnet_amd import (STrajNet, get_pred_waypoint_logits, warpped_gt, compute_occupancy_flow_metrics,
                              OccupancyFlowTaskConfig, apply_sigmoid_to_occupancy_logits)
    m32 = STrajNet(CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.float32, device=dev, seed=0)
    with torch.no_grad():
        m32.flat_weights().copy_(model.flat_weights())

    def fwd(m, xx, training=False):
        return m(xx['ogm'], xx['map_img'], training=training, obs=xx['obs'], occ=xx['occ'], mapt=None, flow=xx['flow'])

    def step32():
        m32.zero_grad()
        out = fwd(m32, x, True)
        d = loss_fn(get_pred_waypoint_logits(out), warpped_gt(x['gt_obs'], x['gt_occ'], x['gt_flow'], x['origin_flow']), None)
        d.total.backward()
    graphed32 = None
    try:                                     # the same capture as the headline step (one hipGraph per step)
        from strajnet_amd.graph import GraphedTrainStep
        graphed32 = GraphedTrainStep(m32, loss_fn, x)
        step = graphed32
    except Exception:
        step = step32
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    dt32 = (time.perf_counter() - t0) / 5
    res = {'parity_mode': {'dtype': 'f32', 'value': round(B / dt32, 2), 'unit': 'scenes/s', 'ms_per_step': round(dt32 * 1e3, 2),
                           'note': 'same train step, f32 storage + exact-f32 MFMA (v_mfma_f32_16x16x4_f32), ' +
                                   ('replayed hipGraph' if graphed32 is not None else 'eager (capture failed)') + ', no optimizer',
                           'max_abs_vs_oracle': None}}
    del graphed32
    with torch.no_grad():
        y16 = fwd(model, x).float()
        y32 = fwd(m32, x).float()
        diff = y16 - y32
        cfgm = OccupancyFlowTaskConfig(256, 256, 8)
        gt = warpped_gt(x['gt_obs'], x['gt_occ'], x['gt_flow'], x['origin_flow'])
        m16 = compute_occupancy_flow_metrics(cfgm, gt, apply_sigmoid_to_occupancy_logits(get_pred_waypoint_logits(y16))).values.tolist()
        mm32 = compute_occupancy_flow_metrics(cfgm, gt, apply_sigmoid_to_occupancy_logits(get_pred_waypoint_logits(y32))).values.tolist()
    res['bf16_error'] = {'vs': 'f32 mode of the same HIP path, same weights and batch, eval forward', 'max_abs': round(float(diff.abs().max()), 5),
                         'rms': round(float(diff.pow(2).mean().sqrt()), 6), 'logit_scale_max_abs': round(float(y32.abs().max()), 3),
                         'dAUC_observed': round(abs(m16[0] - mm32[0]), 6), 'dAUC_occluded': round(abs(m16[1] - mm32[1]), 6),
                         'dAUC_flow_warped': round(abs(m16[5] - mm32[5]), 6)}
    if with_cpu:
        try:
            cb, y_ref, w, xo = cpu_baseline()
            res['cpu_baseline'] = cb
            m32.load_weights(w)
            xt = {k: torch.as_tensor(v).to(dev) for k, v in xo.items()}
            with torch.no_grad():
                y = fwd(m32, xt).cpu().numpy()
            import numpy as np
            res['parity_mode']['max_abs_vs_oracle'] = float(f'{np.abs(y - y_ref).max():.3e}')
            res['parity_mode']['oracle'] = 'oracle/torch_ref.py f32 forward on CPU, B=1 cfg-256, oracle inputs/weights (PARITY UNPINNED by the reference: TensorFlow cannot run here)'
        except Exception as e:
            res['cpu_baseline'] = {'value': None, 'unit': 'scenes/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': f'failed: {e}'}
    return res


def run_extra_configs(steps=30, warmup=10):        # (10 + 3 steps of a fresh process read 4 % low: 4823 vs 5011 scenes/s for config 4 in the same session)
    """BASELINE configs 4 and 5 measured in the SAME default run (the driver only runs `bench.py --gpus 1`): each is this file again,
    in a child process on the same GPU, with its own hipGraph capture, timed region and per-kernel roofline pass.  -> dict for the
    `extra_configs` field of the headline line (never the headline itself)."""
    import subprocess
    res = {}
    for name, flags in (('infer', ['--infer']), ('cfg512', ['--cfg512'])):
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(steps), '--warmup', str(warmup), '--no-cpu-baseline',
               '--no-extra-configs'] + flags
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = [l for l in r.stdout.strip().splitlines() if l.startswith('{')][-1]
            d = json.loads(line)
            roof = d.get('roofline') or {}
            res[name] = {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'], 'warmup': d['warmup'],
                         'dtype': d['dtype'], 'workload': d['config']['workload'], 'global_batch': d['config']['global_batch'],
                         'roofline': {k: roof.get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'mfma_frac_algorithmic', 'avg_launch_ms',
                                                               'launches_per_step', 'mfma_frac_executed', 'hbm_frac', 'serial_kernel_ms_per_step',
                                                               'end_to_end_frac')} if roof else None,
                         'wall_s': round(time.time() - t0, 1)}
        except Exception as e:           # a reported extra, never the headline
            res[name] = {'value': None, 'error': f'{type(e).__name__}: {e}'[:300]}
    return res


def input_feed(graphed, x, steps, finish_step, dev):
    """The step WITH its input feed (reference train.py:85-103,319: every step consumes a new batch): strajnet_amd.data.HostFeed -- the
    batch is uploaded from pinned host memory by a worker thread (1.5 MB pieces on a copy stream) while the previous step runs, then
    lands in the captured step's static inputs.  Two host formats: 'f32' = the decoded float32 tensors; 'raw' = the TFRecord's own bytes
    (bool occupancy grids, int8 map, f32 flows) expanded on the device by stj_decode_raw.  -> {'f32': {...}, 'raw': {...}} scenes/s
    including the feed."""
    import torch
    from strajnet_amd.data import HostFeed
    B = x['ogm'].shape[0]
    raw_kind = {'ogm': 'bool', 'gt_obs': 'bool', 'gt_occ': 'bool', 'map_img': 'int8'}
    res = {}
    for mode in ('f32', 'raw'):
        host, raw = {}, {}
        for k, v in x.items():
            if k not in graphed.static:
                continue
            if mode == 'raw' and k in raw_kind:
                h = (v != 0).to(torch.uint8) if raw_kind[k] == 'bool' else torch.round(v * 256.0).to(torch.int8).view(torch.uint8)
                host[k], raw[k] = h.cpu().contiguous().pin_memory(), raw_kind[k]
            else:
                host[k] = v.detach().float().cpu().contiguous().pin_memory()
        nbytes = sum(h.numel() * h.element_size() for h in host.values())
        feed = HostFeed(graphed.static, host, raw)
        feed.start()
        for _ in range(2):
            feed.land(); graphed(); finish_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            feed.land()                               # this batch -> static inputs; the NEXT batch starts crossing PCIe under this step
            graphed()
            finish_step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        feed.wait_uploaded()
        feed.close()
        res[mode] = {'value': round(B * steps / dt, 3), 'unit': 'scenes/s', 'ms_per_step': round(dt / steps * 1e3, 3),
                     'host_mbytes_per_step': round(nbytes / 1e6, 1)}
    res['note'] = ('same captured step, every step preceded by a fresh batch from pinned host memory (strajnet_amd.data.HostFeed: worker-thread '
                   'upload in 1.5 MB pieces on a copy stream under the previous step, then device copies / stj_decode_raw into the static '
                   'inputs); `value` of the headline has the inputs resident')
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--settle', type=int, default=40, help='untimed set-up replays in front of the warm-up steps (steady-state clocks / caches); 0 = none')
    ap.add_argument('--batch', type=int, default=8, help='scenes per GPU')
    ap.add_argument('--dtype', default=None, choices=['bf16', 'f32', 'f16'], help='default bf16 (train step) / f16 (--infer)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-agent-pipeline', action='store_true', help='inference: the agent branch inside the one captured graph (round 4 form)')
    ap.add_argument('--no-kernel-timing', action='store_true', help='skip every extra pass after the timed region (per-kernel HIP-event timing, the optimizer-free repeat): the run then executes exactly warmup + steps steps (+ 2 capture warm-ups), which is what the rocprofv3 summaries divide by')
    ap.add_argument('--no-graph', action='store_true', help='run the step eagerly instead of replaying the captured hipGraph')
    ap.add_argument('--cfg512', action='store_true', help='BASELINE config 5 instead of the metric config: 512x512 rasters, large_ogm, depths [2,2,6] (extra measurement, not the headline)')
    ap.add_argument('--infer', action='store_true', help='BASELINE config 4 instead of the metric config: inference-only forward, batch 32/GPU, fp16 MFMA path, hipGraph replay (extra measurement, not the headline)')
    ap.add_argument('--no-optimizer', action='store_true', help='time fwd + loss + bwd (+ all-reduce) only, without the fused Keras-Nadam update (train.py:197,224) that the default step ends with')
    ap.add_argument('--no-overlap', action='store_true', help='N>1: one all-reduce of the whole gradient buffer after a monolithic backward instead of two buckets overlapped with the encoder backward')
    ap.add_argument('--serial', action='store_true', help='no side streams: every kernel runs alone (the mode the roofline kernel timings are taken in)')
    ap.add_argument('--no-extra-configs', action='store_true', help='default N=1 run only: skip the child runs of BASELINE configs 4 (--infer) and 5 (--cfg512) that fill `extra_configs`, and the input-feed measurement')
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
    # weights: the same seed on every rank (replicas start identical); Dropout / DropPath stream: per rank, like MirroredStrategy's
    # independent per-replica draws
    model = STrajNet(cfg, fg_msa=True, fg=True, large_ogm=args.cfg512, dtype=dtype, device=dev, seed=0, dropout_seed=rank)
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

    # Data parallel exchange: SUM all-reduce of the flat f32 gradient buffer, as two buckets.  Backward is cut at the
    # raster encoder's outputs; the tail bucket (decoder / attention / trajNet gradients, complete first) is reduced on RCCL's
    # stream UNDER the encoder's backward, the encoder bucket after it.  --no-overlap: one bucket after a monolithic backward.
    from strajnet_amd import dp

    finish_step_ref = [None]

    def build_step(overlap):
        sync = dp.OverlappedGradSync(model) if overlap else None
        model.cut_encoder = overlap

        def finish_step():
            if world > 1:
                if overlap:
                    sync.head_and_wait()
                else:
                    dist.all_reduce(model.flat_grads(), op=dist.ReduceOp.SUM)
            if opt is not None:
                opt.step()
        finish_step_ref[0] = finish_step

        def step():
            model.zero_grad()
            tw = warpped_gt(x['gt_obs'], x['gt_occ'], x['gt_flow'], x['origin_flow'])
            unit = getattr(loss_fn, 'unit_grad', None)       # announced by GraphedTrainStep: the eager passes (per-kernel timing) then run the
            if unit is not None:                             # same one-pass loss the replayed step does
                loss_fn.prepare(tw)
            out = model(x['ogm'], x['map_img'], training=True, obs=x['obs'], occ=x['occ'], mapt=x['mapt'], flow=x['flow'])
            d = loss_fn(get_pred_waypoint_logits(out), tw, None)
            total = d.total                      # observed_xe + occluded_xe + flow + flow_warp_xe (train.py:221)
            if unit is not None:
                total.backward(unit)
            else:
                total.backward()
            if overlap:
                sync.tail()
                model.backward_encoder()
            finish_step()
            return total

        graphed = None
        if not args.no_graph:
            try:
                from strajnet_amd.graph import GraphedTrainStep
                graphed = GraphedTrainStep(model, loss_fn, x, split=overlap)
            except Exception as e:           # capture is an optimisation, never a requirement
                print(f'bench.py: hipGraph capture failed ({type(e).__name__}: {e}); running eagerly', file=sys.stderr)
                graphed = None
                model.cut_encoder = overlap
        eager_step = step

        def step_graph():
            graphed(between=sync.tail if overlap else None)
            finish_step()
            return graphed.total             # the four terms' sum, written by the loss kernel inside the graph (no reduction launch between replays)
        if graphed is not None:
            step = step_graph

        return step, eager_step, graphed, sync

    overlap = world > 1 and not args.no_overlap
    step, eager_step, graphed, sync = build_step(overlap)
    probed = 0
    overlap_fallback_reason = None
    if overlap:          # the two-graph / two-bucket path has only ever run over gloo (no multi-GPU node so far): probe it once, and fall
        err = None       # back to the one-bucket exchange rather than lose the run if RCCL disagrees (the probe is the first warm-up step)
        try:
            step()
            torch.cuda.synchronize()
            probed = 1
        except Exception as e:
            err = f'{type(e).__name__}: {e}'[:300]
        # every rank must take the same path (mismatched collectives hang): agree on the outcome with a MIN all-reduce
        ok = torch.tensor([0.0 if err else 1.0], device=dev)
        try:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            all_ok = bool(ok.item() >= 1.0)
        except Exception as e:
            all_ok, err = False, (err or f'agreement all-reduce failed: {type(e).__name__}: {e}'[:300])
        if not all_ok:
            overlap_fallback_reason = err or 'another rank failed the overlapped probe step'
            print(f'bench.py: overlapped gradient exchange failed ({overlap_fallback_reason}); using one bucket after backward', file=sys.stderr)
            if sync is not None:
                for w in sync._work:         # drain what the failed probe left in flight before the step is rebuilt
                    try:
                        w.wait()
                    except Exception:
                        pass
                sync._work = []
            overlap, probed = False, 0
            step, eager_step, graphed, sync = build_step(False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if os.environ.get('STJ_BENCH_CALIB') == '1':        # PMC passes (tools/pmc_step.sh): a copy of known size calibrates FETCH_SIZE / WRITE_SIZE
        cal = torch.empty(128 * 1024 * 1024, dtype=torch.bfloat16, device=dev).normal_()
        torch.empty_like(cal).copy_(cal)
        del cal
    # Set-up, like the capture warm-ups: `--settle` more replays before the W warm-up steps.  A fresh process measures the first tens of
    # steps 1-3 % low (clocks, caches, allocator: 10 + 3 steps 1330-1370 vs 60 + 10 steps 1386-1392 scenes/s on one box, round 6); the
    # metric is the steady-state rate, so the timed K steps start from a settled device.  Reported as config.settle_steps.
    for _ in range(max(0, args.settle)):
        step()
    for _ in range(max(0, args.warmup - probed)):
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
        allreduce = {'ms': round(ms, 4), 'mbytes': round(nbytes / 1e6, 1),
                     'collective': 'all_reduce(sum, f32) of the whole flat gradient buffer, timed alone (the step itself reduces it as '
                                   + ('two buckets, the tail one under the encoder backward)' if overlap else 'one bucket after backward)'),
                     'overlapped': bool(overlap), 'bucket_mbytes': [round(model.bucket_split * 4 / 1e6, 1), round((g.numel() - model.bucket_split) * 4 / 1e6, 1)],
                     'busbw_GBps': round(2 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9, 1),
                     'frac_of_step': round(ms / (dt_s / args.steps * 1e3), 4)}
    dist_info = None
    if world > 1:
        dist_info = dp.OverlappedGradSync.info()
        dist_info['allreduce_overlapped_with_backward'] = bool(overlap)
        dist_info['overlap_fallback_reason'] = overlap_fallback_reason
        chk = torch.ones(1, device=dev)
        dist.all_reduce(chk)
        dist_info['ranks_seen_by_allreduce'] = int(chk.item())       # every rank contributed 1: proof the collective spans N ranks
    # per-kernel HIP-event timing of EVERY C-ABI launch (strajnet_amd/prof.py): the same launches, issued eagerly with events
    # recorded on the launch stream -- under graph replay individual launches cannot carry events
    prof_ser = prof_conc = None
    nprof = min(args.steps, 3)
    if not args.no_kernel_timing:
        from strajnet_amd import prof as kprof
        # (a) kernels ALONE: side streams off, so an event pair brackets exactly one kernel -- the roofline figures
        model.serial = True
        eager_step()
        kprof.enable()
        for _ in range(nprof):
            eager_step()
        barrier()
        prof_ser = kprof.disable()
        # (b) as scheduled in the measured step (branches on concurrent streams share the GPU): reported next to (a)
        model.serial = args.serial
        eager_step()
        kprof.enable()
        for _ in range(nprof):
            eager_step()
        barrier()
        prof_conc = kprof.disable()
    if world > 1:
        t = torch.tensor([dt_s], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_s = float(t.item())
    loss_val = float(last.detach())
    if world > 1:
        lg = torch.tensor([loss_val], dtype=torch.float64, device=dev)
        dist.all_reduce(lg)                                          # the replica-scaled losses add up to the loss of the global batch
        dist_info['loss_global_batch'] = round(float(lg.item()), 4)
    feed = None
    if rank == 0 and world == 1 and graphed is not None and not args.no_extra_configs and not args.no_kernel_timing:
        try:
            feed = input_feed(graphed, x, args.steps, finish_step_ref[0], dev)
        except Exception as e:
            feed = {'error': f'{type(e).__name__}: {e}'[:300]}
    extras = {}
    if rank == 0 and world == 1 and not args.no_kernel_timing and not args.cfg512 and dtype == torch.bfloat16:
        try:
            extras = parity_extras(model, loss_fn, x, B, dev, not args.no_cpu_baseline)
        except Exception as e:          # reported extras, never the product path
            extras = {'parity_mode': {'error': f'{type(e).__name__}: {e}'}}

    if rank == 0:
        scenes = B * world * args.steps
        algo_step = ALGO_GFLOP_STEP_PER_SCENE_512 if args.cfg512 else ALGO_GFLOP_STEP_PER_SCENE
        value = scenes / dt_s
        peak = PEAK_BF16_TFLOPS if dtype != torch.float32 else PEAK_F32_TFLOPS
        roof, families, kern = build_roofline(prof_ser, prof_conc, nprof, value, B, world, algo_step, peak, args.gemm_trace)
        out = {
            'metric': 'scenes/sec (fwd+bwd, 256x256 grids)', 'value': round(value, 3), 'unit': 'scenes/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt_s / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'STrajNet {"cfg-512 (large_ogm, depths [2,2,6])" if args.cfg512 else "cfg-256"} train step (fwd+OGMFlow_loss+bwd{"+RCCL grad all-reduce" if world > 1 else ""}{"+Nadam" if opt is not None else ""}), '
                                   f'batch {B}/GPU, 8 waypoints, obs+occ+flow heads, fg_msa+fg, random-init weights',
                       'global_batch': B * world, 'grid': '512x512x11 rasters, 256x256 output' if args.cfg512 else '256x256x11', 'parallelism': f'dp{world}', 'hipgraph': graphed is not None, 'concurrent_branch_streams': not args.serial,
                       'optimizer_in_step': opt is not None, 'algorithmic_gflop_per_scene_step': algo_step, 'settle_steps': max(0, args.settle)},
            'loss': round(loss_val, 4),
            'without_optimizer': fwd_bwd_only,
            'allreduce': allreduce,
            'roofline': roof,
        }
        if world > 1:
            out['distributed'] = dist_info
        if families:
            out['families'] = families
            out['kernel_ms'] = kern
        out.update(extras)
        if feed is not None:
            out['with_input_feed'] = feed
        if world == 1 and not args.cfg512 and not args.no_extra_configs and not args.no_kernel_timing and dtype == torch.bfloat16:
            del graphed
            torch.cuda.empty_cache()
            out['extra_configs'] = run_extra_configs()
        if 'cpu_baseline' not in out:
            if not args.no_cpu_baseline and world == 1 and not args.cfg512:
                try:
                    out['cpu_baseline'] = cpu_baseline()[0]
                except Exception as e:      # the CPU port is a reported extra, never the product path
                    out['cpu_baseline'] = {'value': None, 'unit': 'scenes/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': f'failed: {e}'}
            else:
                out['cpu_baseline'] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
