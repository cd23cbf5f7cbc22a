"""Where do the 2-3 ms of the input feed go?  The captured B=8 cfg-256 train step with variants of the per-step feed
(bench.py input_feed): upload only / land only / upload issued after the replay / chunked uploads / kernels reading pinned host memory."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from strajnet_amd import STrajNet, OGMFlow_loss, OccupancyFlowTaskConfig, Nadam, ops
from strajnet_amd.graph import GraphedTrainStep

dev = torch.device('cuda:0')
model = STrajNet(bench.CFG256, fg_msa=True, fg=True, large_ogm=False, dtype=torch.bfloat16, device=dev, seed=0, dropout_seed=0)
loss_fn = OGMFlow_loss(OccupancyFlowTaskConfig(256, 256, 8), ogm_weight=1000.0, occ_weight=1000.0, flow_weight=1.0, replica=1.0,
                       flow_origin_weight=1000.0, no_use_warp=False, use_pred=False, use_focal_loss=False, use_gt=True)
x = bench.synth_batch(8, 1234, dev, 256)
opt = Nadam.for_model(model, lr=1e-4)
graphed = GraphedTrainStep(model, loss_fn, x)
raw_kind = {'ogm': 'bool', 'gt_obs': 'bool', 'gt_occ': 'bool', 'map_img': 'int8'}
KIND = {'bool': 0, 'int8': 1}
host, stage = {}, {}
for k, v in x.items():
    if k not in graphed.static:
        continue
    if k in raw_kind:
        h = (v != 0).to(torch.uint8) if raw_kind[k] == 'bool' else torch.round(v * 256.0).to(torch.int8).view(torch.uint8)
        host[k] = h.cpu().contiguous().pin_memory()
    else:
        host[k] = v.detach().float().cpu().contiguous().pin_memory()
    stage[k] = torch.empty(host[k].shape, dtype=host[k].dtype, device=dev)
nbytes = sum(h.numel() * h.element_size() for h in host.values())
print('host bytes per step %.1f MB' % (nbytes / 1e6), {k: tuple(v.shape) for k, v in host.items()})
copy = torch.cuda.Stream(dev, priority=int(os.environ.get('COPY_PRIO', '0')))
up, landed = torch.cuda.Event(), torch.cuda.Event()


def upload(chunks=1):
    with torch.cuda.stream(copy):
        copy.wait_event(landed)
        for k in host:
            if chunks == 1:
                stage[k].copy_(host[k], non_blocking=True)
            else:
                hs, ss = host[k].view(-1), stage[k].view(-1)
                n = hs.numel(); c = (n + chunks - 1) // chunks
                for i in range(0, n, c):
                    ss[i:i + c].copy_(hs[i:i + c], non_blocking=True)
        up.record(copy)


def land(src=None):
    main = torch.cuda.current_stream(dev)
    if src is None:
        main.wait_event(up)
    for k, st in (src or stage).items():
        dst = graphed.static[k]
        if k in raw_kind:
            n = dst.numel()
            ops.call('stj_decode_raw', ops._p(st), KIND[raw_kind[k]], ops._p(dst), 1, 1, n, 1, 0, 0, 1, n, (1.0 / 256.0) if raw_kind[k] == 'int8' else 1.0, ops._st())
        else:
            dst.copy_(st, non_blocking=True)
    landed.record(main)


def run(name, body, steps=40):
    landed.record(torch.cuda.current_stream(dev)); upload()
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        body()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    print(f'{name:50s} {dt:7.3f} ms/step  {8 / dt * 1e3:8.1f} scenes/s', flush=True)


def v_resident(): graphed(); opt.step()
def v_base(): land(); upload(); graphed(); opt.step()
def v_upload_only(): landed.record(torch.cuda.current_stream(dev)); upload(); graphed(); opt.step()
def v_land_only(): land(stage); graphed(); opt.step()
def v_upload_after(): land(); graphed(); upload(); opt.step()
def v_chunk16(): land(); upload(16); graphed(); opt.step()
def v_hostread(): land(host); graphed(); opt.step()          # kernels / copies read the pinned host buffers directly (zero-copy)
import threading, queue
_q, _done = queue.Queue(), threading.Event()
def _worker():
    torch.cuda.set_device(dev)
    while True:
        item = _q.get()
        if item is None: return
        upload(int(os.environ.get('UP_CHUNKS', '1'))); _done.set()
threading.Thread(target=_worker, daemon=True).start()
_done.set()
def v_threaded():
    _done.wait(); _done.clear()          # the previous upload has been enqueued (its `up` event is recorded)
    land(); _q.put(1); graphed(); opt.step()
only = os.environ.get('FEED_ONLY')
for name, fn in (('resident', v_resident), ('base: land, upload, replay', v_base), ('upload only', v_upload_only), ('land only (stale stage)', v_land_only),
                 ('land, replay, upload', v_upload_after), ('upload in 16 chunks per tensor', v_chunk16), ('land straight from pinned host memory', v_hostread),
                 ('threaded: land, [worker thread: upload], replay', v_threaded), ('resident again', v_resident)):
    if only and not name.startswith(only):
        continue
    try:
        run(name, fn, steps=6 if only else 40)
    except Exception as e:
        print(name, 'FAILED', type(e).__name__, str(e)[:200])
