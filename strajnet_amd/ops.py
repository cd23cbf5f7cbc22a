"""torch.autograd plumbing around the C-ABI HIP kernels.

PyTorch is used for device memory, streams and the autograd tape only; every FLOP-carrying op below is a
launch of a hand-written gfx950 kernel from libstrajnet_hip.so.  Weight gradients are accumulated by the
kernels straight into the model's flat f32 gradient buffer (the DP all-reduce bucket) -- autograd only
routes activation gradients.
"""
import ctypes

import os
import threading

import torch

from . import prof
from ._lib import call as _raw_call


def call(name, *args):
    """One C-ABI call (strajnet_amd._lib.call); bracketed by HIP events when bench.py's per-kernel timing pass is on."""
    if prof.ACTIVE is None:
        return _raw_call(name, *args)
    prof.record(name, args, lambda: _raw_call(name, *args))


GROUP_GEMMS = os.environ.get('STJ_GEMM_GROUP', '1') != '0'
class _GroupState(threading.local):
    """per host thread (the forward thread and autograd's device threads each record into their own group, as the C ABI's contract says)"""
    def __init__(self):
        self.h = [None, None]       # [the open group's handle (None: stj_gemm launches at once), its host buffer]
        self.depth = [0]


_GS = _GroupState()


class _GroupProxy:
    """_GROUP[i] / _GROUP_DEPTH[i] of the calling thread"""
    def __init__(self, attr):
        self.attr = attr

    def __getitem__(self, i):
        return getattr(_GS, self.attr)[i]

    def __setitem__(self, i, v):
        getattr(_GS, self.attr)[i] = v


_GROUP = _GroupProxy('h')
# Layers with at least this many rows launch their input and weight gradient separately: each then runs with 192-element k-tiles
# (gemm_deepk_kernel: a third of the barrier-bound links), which the grouped kernel does not have.  Measured, scenes/s: no grouping
# 943, limit 1024 rows 939, limit 16384 rows (every small layer grouped) 927.
_GROUP_MAX_ROWS = 8192      # re-measured at the end of round 2: 1024 -> 1071, 4096 -> 1082, 8192 -> 1088, 16384 -> 1078, 65536 -> 1064 scenes/s
_GROUP_DEPTH = _GroupProxy('depth')


class gemm_group:
    """`with gemm_group():` -- the stj_gemm calls inside (independent of each other, same stream) are recorded and leave as one
    launch (stj_gemm_group_begin / _end).  A no-op while bench.py's per-kernel timing pass is on, and when nested."""
    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        self.on = self.enabled and GROUP_GEMMS and prof.ACTIVE is None and _GROUP_DEPTH[0] == 0
        _GROUP_DEPTH[0] += 1
        if self.on:
            if _GROUP[1] is None:       # caller-owned host memory the C ABI records the group in (the library keeps no state)
                from ._lib import lib
                _GROUP[1] = ctypes.create_string_buffer(int(lib().stj_gemm_group_workspace_bytes()))
            h = ctypes.cast(_GROUP[1], vp)
            _raw_call('stj_gemm_group_begin', h)
            _GROUP[0] = h
        return self

    def __exit__(self, et, ev, tb):
        _GROUP_DEPTH[0] -= 1
        if self.on:
            h, _GROUP[0] = _GROUP[0], None
            _raw_call('stj_gemm_group_end', h, _st())
        return False


ACT_NONE, ACT_GELU, ACT_ELU = 0, 1, 2
U_GELU, U_ELU, U_TANHS = 1, 2, 3
vp = ctypes.c_void_p


DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}      # STJ_F32 / STJ_BF16 / STJ_F16 (strajnet_hip.h)


def _dt(t):
    try:
        return DTYPE_CODE[t.dtype]
    except KeyError:
        raise TypeError(f'unsupported activation dtype {t.dtype}') from None


def _p(t):
    if t is None:
        return vp(0)
    if isinstance(t, vp):
        return t
    return vp(t.data_ptr())


def _poff(t, elems):
    """raw pointer `elems` elements past the start of tensor t (may point outside t's own view: flat-buffer addressing)."""
    return vp(t.data_ptr() + elems * t.element_size())


def _st():
    return vp(torch.cuda.current_stream().cuda_stream)


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('strajnet_amd ops run on the GPU only (HIP kernels); got a CPU tensor. '
                               'There is no CPU fallback.')


_WS = {}


def _workspace(device, size_fn, which=0):
    """Caller-owned kernel scratch, allocated once per (device, kernel family, slot) and ZEROED once (arrival counters re-arm
    themselves); launches on one stream reuse it in stream order."""
    key = (str(device), size_fn, which)
    ws = _WS.get(key)
    if ws is None:
        from ._lib import lib
        ws = _WS[key] = torch.zeros(int(getattr(lib(), size_fn)()), dtype=torch.uint8, device=device)
    return ws


# ----------------------------------------------------------------------------------------------------
# Zeroed scratch.  A step needs ~20 zero-initialised f32 buffers (folded conv tap gradients, batched weight-gradient staging,
# atomically accumulated sums); as separate torch.zeros calls each is a ~5 us fill launch.  They are carved out of ONE arena that
# the model re-zeroes (used prefix only) in zero_grad(): one fill per step.  Regions stay valid until the next zero_grad(); callers
# that never arm the arena (op-level tests) get plain torch.zeros.
# ----------------------------------------------------------------------------------------------------
class _ZeroArena:
    FLOATS = 8 << 20           # 32 MB: ~2x what a cfg-512 step takes
    # floats; every region starts on a multiple of it.  256 bytes: the f32 atomics of the decoder weight-gradient kernels retire 15-40 %
    # slower into a buffer that starts on an odd multiple of 32 bytes (tools/probes/wgrad_align.py: 105 -> 147 us for 192 -> 128,
    # 213 -> 243 us for 96 -> 48; any multiple of 64 bytes is as good as 4 KB)
    ALIGN = 64

    def __init__(self):
        self.buf, self.off, self.armed = None, 0, False
        self.split_ws = {}        # the owner's Swin split workspaces (_swin_ws)
        self.split_ws_stream = {}     # ... the stream that used each last, and whether that use was inside a graph capture
        self.split_ws_captured = {}

    def _here(self, device):
        d = torch.device(device)
        return self.buf is not None and self.buf.device.type == d.type and (d.index is None or d.index == self.buf.device.index)

    def adopt(self, buf):
        """Use `buf` (zeroed f32 storage the owner re-zeroes itself: see rearm) instead of an allocation of the arena's own."""
        self.buf, self.off, self.armed, self.external = buf, 0, False, True

    def rearm(self):
        """Start of a step for an adopted buffer: the OWNER has just zeroed buf[:off] (one fill together with its own tensors)."""
        self.off, self.armed = 0, True

    def arm(self, device):
        """Start of a step: everything handed out so far is dead; zero it again."""
        if not self._here(device):
            self.buf = torch.zeros(self.FLOATS, dtype=torch.float32, device=device)
        elif self.off:
            self.buf[:self.off].zero_()
        self.off, self.armed = 0, True

    def take(self, n, device):
        A = self.ALIGN
        n8 = (n + A - 1) // A * A                 # keep every region aligned (>= 32 bytes)
        if not self.armed or not self._here(device) or self.off + n8 > self.buf.numel():
            return None
        v = self.buf[self.off:self.off + n]
        self.off += n8
        return v


_ARENA = _ZeroArena()        # the arena of the model whose step is being enqueued (each STrajNet owns one; see use_arena)


def use_arena(arena):
    """Make `arena` the one zeros_f32 carves from.  A model calls this from zero_grad() and call(): regions it took stay valid
    until ITS next zero_grad(), whatever other models on the device do in between."""
    global _ARENA
    _ARENA = arena


def zeros_f32(shape, device):
    """Zero-initialised f32 scratch of `shape`, valid until the model's next zero_grad()."""
    n = 1
    for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
        n *= int(d)
    v = _ARENA.take(n, device)
    if v is None:
        return torch.zeros(shape, dtype=torch.float32, device=device)
    return v.view(shape)


# ----------------------------------------------------------------------------------------------------
# Weight gradients on a side stream.  In backward the data-gradient chain (dY -> dX -> ...) is the critical path; the weight
# gradient of a layer (x^T dY, accumulated into the flat gradient buffer) is a leaf of the dependency graph.  Most kernels of
# this model are latency-bound and leave CUs idle, so the wgrad launches go to a second stream that forks from the current one
# (dY is complete) and is joined ONCE, when the autograd engine finishes the backward pass (queue_callback).  Operands stay
# referenced until the join, so the caching allocator cannot hand their memory to later kernels of the main stream.
# ----------------------------------------------------------------------------------------------------
_ROLE_STREAMS = {}


def role_stream(device, role, priority=0):
    """ONE stream per (device, role) for the whole process.  torch hands out side streams from a round-robin pool of 32 per device: a
    process that builds many models / captured steps (the GPU test suite: a model per test) wraps the pool, and a new model's side stream
    is then the SAME stream as some long-lived one (the capture stream, the weight-gradient stream).  And the number of DISTINCT streams a
    process uses matters on ROCm 7.2: with one more side stream alive (a sixth: the inference agent pipeline's own, round 5) a later,
    unrelated hipGraph replay died in hip::Graph::UpdateStreams (tests/test_model_gpu.py run as a whole; gone when that work moved onto an
    existing stream).  Models on one device share their side streams: at worst a false dependency between two models."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _ROLE_STREAMS.get((idx, role))
    if st is None:
        st = _ROLE_STREAMS[(idx, role)] = torch.cuda.Stream(torch.device('cuda', idx), priority=priority)
    return st


_WG = {}
# bit 0: dense layers, bit 1: up-convs.  Measured (B=8): up-convs on the side stream +1 %; dense layers -9 % (their 768-block
# split-K kernels crowd the data-gradient chain out of the CUs), so only the up-convs use it by default.
_WG_MODE = 2


_SERIAL = False


def set_serial(flag):
    """True: weight-gradient launches stay on the current stream (per-kernel timing with HIP events needs kernels to run alone)."""
    global _SERIAL
    _SERIAL = bool(flag)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _wgrad_join(key):
    st = _WG[key]
    st['main'].wait_stream(st['side'])
    st['keep'].clear()
    st['armed'] = False


def wgrad_join_now(main):
    """Order `main` after everything queued on the weight-gradient side stream of main's device (used where a backward pass is
    followed by work the autograd engine's callbacks do not cover)."""
    key = main.device.index if main.device.index is not None else torch.cuda.current_device()
    st = _WG.get(key)
    if st is not None:
        main.wait_stream(st['side'])


def wgrad_stream(kind, *operands):
    """Context manager: kernels launched inside run on the weight-gradient side stream of the operands' device."""
    if _SERIAL or not (_WG_MODE & kind):
        return _NullCtx()
    dev = operands[0].device
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _WG.get(key)
    if st is None:
        # (a low-priority side stream measured nothing: 1341 / 1360 / 1345 vs 1350 / 1363 / 1349 scenes/s)
        st = _WG[key] = {'side': role_stream(dev, 'wgrad'), 'keep': [], 'armed': False, 'main': None}
    main = torch.cuda.current_stream(dev)
    st['side'].wait_stream(main)
    st['keep'].extend(operands)
    if not st['armed']:
        st['armed'], st['main'] = True, main
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _wgrad_join(key))
    return torch.cuda.stream(st['side'])


class _JoinAfterBackward(torch.autograd.Function):
    """Identity on the model output.  Its backward is the FIRST node the engine runs; it queues a callback that makes the
    stream backward was called from wait for the model's side streams once the whole pass has been enqueued -- gradients
    written by side-stream kernels straight into the flat buffer (no AccumulateGrad node the engine could track) are then
    ordered before whatever the caller enqueues next (optimizer, all-reduce, end of a graph capture)."""
    @staticmethod
    def forward(ctx, out, streams, post):
        ctx.streams, ctx.post = streams, post
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        streams, post = ctx.streams, ctx.post
        main = torch.cuda.current_stream(g.device)
        wgrad_queue_begin(g.device)

        def join():
            for s in streams:
                main.wait_stream(s)
            with torch.cuda.stream(main), torch.no_grad():
                wgrad_queue_end(g.device)     # the dense weight gradients still queued (one grouped stream-K launch)
            # the weight-gradient side stream writes bias-gradient partials that `post` folds: it must be ordered before the fold
            # (its own join callback is queued later than this one and would run after it)
            key = g.device.index if g.device.index is not None else torch.cuda.current_device()
            st = _WG.get(key)
            flush_upconv_wgrads()             # (leftovers: a flush point whose backward did not run)
            with torch.cuda.stream(main), torch.no_grad():
                run_pending_wg(g.device)      # (handed-over weight gradients nobody took)
            st = _WG.get(key)
            if st is not None and st['armed']:
                main.wait_stream(st['side'])
            if post is not None:          # e.g. fold the partial-gradient copies into the flat gradient buffer
                with torch.cuda.stream(main), torch.no_grad():
                    post()
        torch.autograd.Variable._execution_engine.queue_callback(join)
        return g, None, None


def join_after_backward(out, streams, post=None):
    streams = [s for s in streams if s is not None]
    if (not streams and post is None) or not out.requires_grad:
        return out
    return _JoinAfterBackward.apply(out, streams, post)


class Param:
    """One trainable tensor: f32 master view, compute-dtype view, f32 gradient view (all slices of flat buffers)."""
    __slots__ = ('name', 'shape', 'master', 'c', 'grad', 'part')

    def __init__(self, name, shape, master, c, grad):
        self.name, self.shape, self.master, self.c, self.grad = name, tuple(shape), master, c, grad
        self.part = None        # (view of copy 0, copies, stride): partial-gradient copies that kernels with many workgroups
                                # per parameter element rotate their atomics over (folded into .grad after backward)


def gemm(A, B, C, M, N, K, sA, sB, sC, dt, bias=None, sBias=(0, 0), res=None, sRes=(0, 0, 0), nb=(1, 1),
         act=ACT_NONE, alpha=1.0, c_f32=0, accumulate=0, splitk=1, colsum=None, kseg=(1, 0, 0), post=None):
    """sA = (b1, b2, m, k) element strides; sB = (b1, b2, k, n); sC = (b1, b2, ldc); sRes = (b1, b2, ld);
    kseg = (n, sA, sB): the contraction also runs over n K-segments of A / B that lie sA / sB elements apart.
    post: callable that consumes C; it travels with a DEFERRED weight gradient and runs behind the flush that carries it.
    Returns True when the product was queued (post will run), False when it was launched (or recorded in the open group): the
    caller then runs its post-processing itself, after closing the group."""
    if (accumulate and isinstance(A, torch.Tensor) and isinstance(B, torch.Tensor) and _wq(A.device)['on'] and splitk == 0 and c_f32
            and kseg[0] == 1 and sA[2] == 1 and sB[3] == 1 and bias is None and res is None and alpha == 1.0):
        # a weight gradient dW += x^T dY inside a model's backward pass: queued for the grouped stream-K launch of the next flush
        j = WJob(A, B, C, colsum, K, M, N, sA[3], sB[2], sC[2], dt, nb=nb, sx=sA[:2], sdy=sB[:2], sdw=sC[:2], sdb=sBias)
        if j.supported():
            wgrad_queue_push(j, post, A.device)
            return True
    _gemm_call(A, B, C, M, N, K, sA, sB, sC, dt, bias, sBias, res, sRes, nb, act, alpha, c_f32, accumulate, splitk, colsum, kseg)
    return False


def _gemm_call(A, B, C, M, N, K, sA, sB, sC, dt, bias, sBias, res, sRes, nb, act, alpha, c_f32, accumulate, splitk, colsum, kseg):
    call('stj_gemm', _p(A), _p(B), _p(C), _p(bias), _p(res), _p(colsum), M, N, K, nb[0], nb[1],
         sA[0], sA[1], sA[2], sA[3], sB[0], sB[1], sB[2], sB[3], sC[0], sC[1], sC[2],
         sBias[0], sBias[1], sRes[0], sRes[1], sRes[2], act, float(alpha), dt, c_f32, accumulate, splitk,
         kseg[0], kseg[1], kseg[2], _GROUP[0], _st())


# ----------------------------------------------------------------------------------------------------
# Grouped stream-K weight gradients (csrc/wgrad_sk.hip): dW += x^T dY (+ db += 1^T dY) of MANY Dense layers in one launch
# ----------------------------------------------------------------------------------------------------
class WJob:
    """One weight-gradient problem: dw[z] += x[z]^T dy[z], db[z] += colsum(dy[z]) over nb = (nb1, nb2) batch levels.
    x / dy / dw / db are tensors or raw device pointers (ctypes.c_void_p) -- the batch strides may reach outside a tensor's own view
    (flat-buffer addressing); sx / sdy / sdw / sdb = (stride of level 1, of level 2) in elements."""
    __slots__ = ('x', 'dy', 'dw', 'db', 'rows', 'cin', 'cout', 'ldx', 'lddy', 'lddw', 'nb', 'sx', 'sdy', 'sdw', 'sdb', 'dt', 'keep')

    def __init__(self, x, dy, dw, db, rows, cin, cout, ldx, lddy, lddw, dt, nb=(1, 1), sx=(0, 0), sdy=(0, 0), sdw=(0, 0), sdb=(0, 0), keep=()):
        self.x, self.dy, self.dw, self.db = x, dy, dw, db
        self.rows, self.cin, self.cout, self.ldx, self.lddy, self.lddw = rows, cin, cout, ldx, lddy, lddw
        self.nb, self.sx, self.sdy, self.sdw, self.sdb, self.dt, self.keep = nb, sx, sdy, sdw, sdb, dt, keep

    def c(self):
        from ._lib import WgradJob
        return WgradJob(_p(self.x), _p(self.dy), _p(self.dw), _p(self.db), self.rows, self.cin, self.cout, self.nb[0], self.nb[1],
                        self.ldx, self.lddy, self.lddw, self.sx[0], self.sx[1], self.sdy[0], self.sdy[1], self.sdw[0], self.sdw[1],
                        self.sdb[0], self.sdb[1])

    def supported(self):
        from ._lib import lib
        return bool(lib().stj_wgrad_job_supported(ctypes.byref(self.c()), self.dt))

    def gemm(self):
        """the same problem through stj_gemm (split-K tile kernel): shapes the stream-K kernel does not take"""
        call('stj_gemm', _p(self.x), _p(self.dy), _p(self.dw), vp(0), vp(0), _p(self.db), self.cin, self.cout, self.rows, self.nb[0], self.nb[1],
             self.sx[0], self.sx[1], 1, self.ldx, self.sdy[0], self.sdy[1], self.lddy, 1, self.sdw[0], self.sdw[1], self.lddw,
             self.sdb[0], self.sdb[1], 0, 0, 0, ACT_NONE, 1.0, self.dt, 1, 1, 0, 1, 0, 0, _GROUP[0], _st())


WG_BUDGET = 0       # workgroups of a grouped weight-gradient launch (0: one per CU; measured 128 / 192: 2 % / 0.5 % slower end to end)
_WJ_LAST = [None]


def _wgrad_group_model(a):
    jobs = _WJ_LAST[0]
    fl = sum(2.0 * j.rows * j.cin * j.cout * j.nb[0] * j.nb[1] for j in jobs)
    by = 0.0
    for j in jobs:
        nb, es = j.nb[0] * j.nb[1], 2
        nx = (j.nb[0] if j.sx[0] else 1) * (j.nb[1] if j.sx[1] else 1)
        by += es * j.rows * (j.cin * nx + j.cout * nb) + 4 * j.cin * j.cout * nb
    return f'wgrad_group[{len(jobs)} jobs, {sum(j.rows * j.nb[0] * j.nb[1] for j in jobs)} rows]', 'gemm_wgrad', fl, fl, by


prof.EXTRA_MODELS['stj_wgrad_group'] = _wgrad_group_model


def wgrad_group(jobs, budget=None):
    """Launch the weight gradients `jobs` (list of WJob, one dtype): those the stream-K kernel takes leave as ONE launch, the rest as
    stj_gemm split-K launches."""
    from ._lib import WgradJob
    fast = [j for j in jobs if j.supported()]
    for j in jobs:
        if j not in fast:
            j.gemm()
    if fast:
        arr = (WgradJob * len(fast))(*[j.c() for j in fast])
        _WJ_LAST[0] = fast
        call('stj_wgrad_group', ctypes.cast(arr, vp), len(fast), fast[0].dt, WG_BUDGET if budget is None else budget, _st())


# Deferred weight gradients.  Inside a model's backward pass (between _JoinAfterBackward.backward, the first node the engine runs, and its
# end-of-pass callback) every dW += x^T dY that gemm() is asked for is QUEUED instead of launched; a flush (the model's flush points and
# the end of the pass) sends everything queued so far as one stj_wgrad_group launch per 28 problems.  The queue holds the operands, so
# the caching allocator cannot recycle them before the flush is enqueued; operands produced on another stream are ordered in front of
# the flush with an event and recorded on the flush stream.
WGRAD_SK = os.environ.get('STJ_WGRAD_SK', '1') != '0'
class _WQueues(dict):
    """One queue per device (the autograd engine runs a device's backward nodes on that device's worker thread with the device
    current): two models stepping on two GPUs of one process keep their queued jobs apart."""
    def __missing__(self, dev):
        q = self[dev] = {'on': False, 'jobs': []}
        return q


_WQS = _WQueues()


def _wq(dev=None):
    """The queue of `dev` (a torch.device / tensor device; None: the calling thread's current device).  Callers that know their model's or
    tensor's device pass it: the forward thread's current device need not be the model's (several GPUs driven by one process), while
    autograd's worker thread of a device always has that device current."""
    if dev is not None and getattr(dev, 'type', 'cuda') == 'cuda' and getattr(dev, 'index', None) is not None:
        return _WQS[dev.index]
    return _WQS[torch.cuda.current_device() if torch.cuda.is_available() else -1]


def wgrad_queue_begin(dev=None):
    q = _wq(dev)
    assert not q['jobs'], 'wgrad_queue_begin: jobs of another backward pass are still queued on this device'
    q['on'] = WGRAD_SK


def wgrad_queue_reset(dev=None):
    """Start of a forward pass: whatever an aborted backward pass left behind is dropped."""
    q = _wq(dev)
    q['on'] = False
    q['jobs'] = []


def wgrad_queue_push(job, post=None, dev=None):
    st = torch.cuda.current_stream(dev)
    ev = None
    if not _SERIAL:
        ev = torch.cuda.Event()
        ev.record(st)
    _wq(dev)['jobs'].append((job, post, st, ev))


def wgrad_queue_flush(dev=None, side=False):
    """Launch everything queued (on `dev`; None: the current device), on the current stream.  (Round 5: the flush-point launches on the weight-gradient side stream instead,
    so that the next stage's backward need not wait for them, measured 1286-1291 scenes/s with all CUs as the launch's budget, 1270-1276
    with 128 workgroups, 1234-1246 with 96, against 1296-1305 on the main stream: the Swin backward kernels they would run beside fill
    the CUs they are given, and the join before the optimizer waits for the slowed-down last flush.)
    side=True: this one launch on the weight-gradient side stream (ordered behind the current stream, joined at the end of the pass)."""
    q = _wq(dev)
    items, q['jobs'] = q['jobs'], []
    if not items:
        return
    if side and not _SERIAL:
        keep = [t for job, _, _, _ in items for t in (job.x, job.dy) if isinstance(t, torch.Tensor)]
        with wgrad_stream(1, *keep):
            _wgrad_launch(items, dev)
    else:
        _wgrad_launch(items, dev)


def _wgrad_launch(items, dev):
    cur = torch.cuda.current_stream(dev)
    last = {}
    for job, post, st, ev in items:
        if ev is not None and st != cur:
            last[st] = ev                       # events of one stream are ordered: the last one covers the earlier ones
            for t in (job.x, job.dy):
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)
    for ev in last.values():
        cur.wait_event(ev)
    wgrad_group([it[0] for it in items])
    for _, post, _, _ in items:
        if post is not None:
            post()


def wgrad_queue_end(dev=None):
    wgrad_queue_flush(dev)
    _wq(dev)['on'] = False


class _WgradQueueFlush(torch.autograd.Function):
    """Identity whose backward flushes the weight-gradient queue: everything downstream of it in the forward pass has been through."""
    @staticmethod
    def forward(ctx, x, side):
        ctx.side = side
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        wgrad_queue_flush(g.device, side=ctx.side)
        return g, None


def wgrad_queue_flush_point(x, side=False):
    if WGRAD_SK and WGRAD_SK_POINTS and x.requires_grad and torch.is_grad_enabled():
        return _WgradQueueFlush.apply(x, side)
    return x


WGRAD_SK_POINTS = True       # flush at the stage boundaries, not only at the end of the pass (end only: 1.5 % slower end to end)


def _splitk(M_out, N_out, Kdim):
    tiles = ((M_out + 63) // 64) * ((N_out + 63) // 64)
    want = max(1, 512 // max(tiles, 1))
    return int(max(1, min(want, (Kdim + 255) // 256, 256)))


# ----------------------------------------------------------------------------------------------------
# Dense (+bias, +ELU, +residual)
# ----------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """y = act(x @ wc + bias) [+ res].  wc: [K,N] compute-dtype weight; gw: f32 [K,N] gradient accumulation target
    (or, with `fold`, a scratch gradient is produced and handed to fold(g))."""
    @staticmethod
    def forward(ctx, x, trig, wc, gw, bias, gb, act, res, fold):
        _req_cuda(x)
        K, N = wc.shape
        x2 = x.contiguous().view(-1, K)
        M = x2.shape[0]
        dt = _dt(x2)
        if act == ACT_ELU and res is not None:
            raise RuntimeError('linear: ELU + residual in one op is not supported')
        if act == ACT_GELU:
            raise RuntimeError('linear: fused GELU has no backward (use the separate gelu op)')
        y = torch.empty((M, N), dtype=x2.dtype, device=x2.device)
        r2 = res.contiguous().view(M, N) if res is not None else None
        gemm(x2, wc, y, M, N, K, (0, 0, K, 1), (0, 0, N, 1), (0, 0, N), dt, bias=bias, res=r2, sRes=(0, 0, N), act=act)
        ctx.wc, ctx.gw, ctx.gb, ctx.act, ctx.has_res, ctx.fold = wc, gw, gb, act, res is not None, fold
        ctx.xshape = x.shape
        ctx.save_for_backward(x2, y if act == ACT_ELU else None)
        return y.view(x.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        x2, y = ctx.saved_tensors
        wc = ctx.wc
        K, N = wc.shape
        M = x2.shape[0]
        dt = _dt(x2)
        dy2 = dy.contiguous().view(M, N)
        if ctx.act == ACT_ELU:
            dpre = torch.empty_like(dy2)
            call('stj_unary_bwd', _p(dy2), _p(y), _p(dpre), M * N, U_ELU, 0.0, dt, _st())
        else:
            dpre = dy2
        dx = None
        gw = ctx.gw if ctx.fold is None else zeros_f32((K, N), x2.device)
        # the two products are independent: one grouped launch for the small layers (big ones keep the row-streaming dgrad kernel)
        with gemm_group(M < _GROUP_MAX_ROWS and not (_WG_MODE & 1)):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x2)
                gemm(dpre, wc, dx, M, K, N, (0, 0, N, 1), (0, 0, 1, N), (0, 0, K), dt)        # dx = dpre W^T
                dx = dx.view(ctx.xshape)
            fold = ctx.fold
            with wgrad_stream(1, x2, dpre):
                queued = gemm(x2, dpre, gw, K, N, M, (0, 0, 1, K), (0, 0, N, 1), (0, 0, N), dt, c_f32=1, accumulate=1,
                              splitk=0, colsum=ctx.gb, post=(lambda: fold(gw)) if fold is not None else None)   # dW += x^T dpre ; db += 1^T dpre (fused)
        if fold is not None and not queued:
            with wgrad_stream(1, x2, dpre):
                fold(gw)
        dres = dy if ctx.has_res else None
        return dx, None, None, None, None, None, None, dres, None


def linear(x, pw, pb=None, act=ACT_NONE, res=None):
    """Keras Dense / 1x1 Conv2D / flattened conv kernel: weight [..., K, N] viewed as [prod(...)*K, N]."""
    N = pw.c.shape[-1]
    return _Linear.apply(x, pw.master, pw.c.view(-1, N), pw.grad.view(-1, N), pb.master if pb is not None else None,
                         pb.grad if pb is not None else None, act, res, None)


# Weight-gradient launches handed from one backward node to another that runs on a different stream (the fused cross-attention's four
# products -> the key / value projections' backward on the agent branch's stream): (closure, event behind the producer, operands).
_PENDING_WG = {}


def run_pending_wg(dev=None):
    """Launch the handed-over weight gradients on the CURRENT stream of `dev` (ordered behind their producers)."""
    idx = dev.index if dev is not None and dev.index is not None else torch.cuda.current_device()
    items = _PENDING_WG.pop(idx, [])
    if not items:
        return
    cur = torch.cuda.current_stream(dev)
    for wg, ev, operands in items:
        cur.wait_event(ev)
        for t in operands:
            t.record_stream(cur)
        wg()


class _HeadsIn(torch.autograd.Function):
    """tfa-MHA query / key / value projection for Z weight sets at once: y[z, r, h*hs + o] = sum_i x[(z|shared), r, i] W[z][h, i, o].
    The kernels [H, in, hs] are read and their gradients written IN PLACE in the flat buffers: every product is a batched GEMM over
    (z, h) with the head as a batch stride (no permuted weight copy per step, no gradient fold afterwards)."""
    @staticmethod
    def forward(ctx, x, trig, w0, gw0, zstride, Z, shared_x):
        _req_cuda(x)
        H, I, hs = w0.shape
        x = x.contiguous()
        R = x.numel() // I if shared_x else x.numel() // (I * Z)
        dt = _dt(x)
        y = torch.empty((Z, R, H * hs), dtype=x.dtype, device=x.device)
        gemm(x, w0, y, R, hs, I, (0 if shared_x else R * I, 0, I, 1), (zstride, I * hs, hs, 1), (R * H * hs, hs, H * hs), dt, nb=(Z, H))
        ctx.dims = (Z, R, H, I, hs, zstride, shared_x, x.shape)
        ctx.w0, ctx.gw0 = w0, gw0
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        Z, R, H, I, hs, zstride, shared_x, xshape = ctx.dims
        dt = _dt(x)
        dy = dy.contiguous()
        run_pending_wg(dy.device)
        dx = acc = None
        with gemm_group(R < _GROUP_MAX_ROWS and not (_WG_MODE & 1)):          # input and weight gradient: independent products
            if ctx.needs_input_grad[0]:
                # dx[z] = sum_h dy[z][:, h-slice] W[z,h]^T : the head sum is the K-segment loop of ONE GEMM per z
                dx = torch.empty(xshape, dtype=x.dtype, device=x.device)
                if shared_x:
                    acc = zeros_f32((R, I), x.device)         # the Z sets accumulate with f32 atomics
                    gemm(dy, ctx.w0, acc, R, I, hs, (R * H * hs, 0, H * hs, 1), (zstride, 0, 1, hs), (0, 0, I), dt, nb=(Z, 1), c_f32=1,
                         accumulate=1, kseg=(H, hs, I * hs))
                else:
                    gemm(dy, ctx.w0, dx, R, I, hs, (R * H * hs, 0, H * hs, 1), (zstride, 0, 1, hs), (R * I, 0, I), dt, nb=(Z, 1),
                         kseg=(H, hs, I * hs))
            with wgrad_stream(1, x, dy):      # dW[z,h] += x_z^T dy_z[:, h-slice], straight into the flat gradient buffer
                gemm(x, dy, ctx.gw0, I, hs, R, (0 if shared_x else R * I, 0, 1, I), (R * H * hs, hs, H * hs, 1), (zstride, I * hs, hs),
                     dt, nb=(Z, H), c_f32=1, accumulate=1, splitk=0)
        if acc is not None:
            call('stj_cast', _p(acc), 0, _p(dx), dt, R * I, _st())
        return (dx,) + (None,) * 6


def linear_heads_in(x, pw):
    """tfa-MHA query/key/value kernel [H, in, hs]: y[..., h*hs+o] = sum_i x[..., i] W[h,i,o] (no bias)."""
    H, I, hs = pw.c.shape
    y = _HeadsIn.apply(x, pw.master, pw.c, pw.grad, 0, 1, False)
    return y.view(x.shape[:-1] + (H * hs,))


def linear_heads_in_z(x, trig, w0, gw0, zstride, Z, shared_x):
    """the same for Z weight sets lying zstride elements apart in the flat buffers; x [Z,R,in] or shared [R,in] -> [Z,R,H*hs]."""
    return _HeadsIn.apply(x, trig, w0, gw0, zstride, Z, shared_x)


def linear_heads_out(x, pw, pb):
    """tfa-MHA projection kernel [H, hs, out] (+ projection_bias): plain dense on the flattened (h,hs) axis."""
    H, hs, O = pw.c.shape
    return _Linear.apply(x, pw.master, pw.c.view(H * hs, O), pw.grad.view(H * hs, O), pb.master, pb.grad, ACT_NONE, None, None)


# ----------------------------------------------------------------------------------------------------
# batched dense over a LEADING "z" axis (8 waypoints / 8 time steps), one launch for all z:
#   y[z, r, :] = act(x[(z|shared), r, :] @ W_z + bias_z)          x: [Z, R, K] or shared [R, K];  y: [Z, R, N]
# z-major layout makes every product a plain batched GEMM (rows (scene, token) of one z are contiguous): forward,
# dgrad and wgrad are ONE launch each.  W_z lives at w0 + z*wstride (elements): the 8 per-waypoint weight sets are
# addressed in place inside the flat parameter buffer (constant stride), and so are their gradients.
# ----------------------------------------------------------------------------------------------------
class _LinearZ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, trig, w0, wstride, b0, bstride, gw0, gwstride, gb0, Z, act, shared_x, fold, grad_is_pre=False):
        _req_cuda(x)
        K, N = w0.shape
        x = x.contiguous()
        ctx.grad_is_pre = grad_is_pre
        R = x.numel() // K if shared_x else x.numel() // (K * Z)
        dt = _dt(x)
        y = torch.empty((Z, R, N), dtype=x.dtype, device=x.device)
        gemm(x, w0, y, R, N, K, (0, 0 if shared_x else R * K, K, 1), (0, wstride, N, 1), (0, R * N, N), dt, bias=b0,
             sBias=(0, bstride), nb=(1, Z), act=act)
        ctx.dims = (Z, R, K, N, shared_x, act, wstride, bstride, gwstride, x.shape)
        ctx.w0, ctx.gw0, ctx.gb0, ctx.fold = w0, gw0, gb0, fold
        ctx.save_for_backward(x, y if act == ACT_ELU else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        Z, R, K, N, shared_x, act, wstride, bstride, gwstride, xshape = ctx.dims
        dt = _dt(x)
        dy = dy.contiguous()
        if act == ACT_ELU and not ctx.grad_is_pre:      # (grad_is_pre: the consumer folded ELU'(y) into the gradient it returned -- upconv_add(skips_pre=True))
            dpre = torch.empty_like(dy)
            call('stj_unary_bwd', _p(dy), _p(y), _p(dpre), dy.numel(), U_ELU, 0.0, dt, _st())
        else:
            dpre = dy
        dx = acc = None
        with gemm_group(R < _GROUP_MAX_ROWS and not (_WG_MODE & 1)):          # input and weight gradient: independent products
            if ctx.needs_input_grad[0]:
                if shared_x and R >= 4096:
                    # dx[r,:] = sum_z dpre[z,r,:] W_z^T : ONE GEMM whose contraction runs over the Z segments (z, n) -- no atomics
                    dx = torch.empty(xshape, dtype=x.dtype, device=x.device)
                    gemm(dpre, ctx.w0, dx, R, K, N, (0, 0, N, 1), (0, 0, 1, N), (0, 0, K), dt, kseg=(Z, R * N, wstride))
                elif shared_x:     # few rows: Z independent launches-in-one fill the GPU better; all z accumulate with f32 atomics
                    acc = zeros_f32((R, K), x.device)
                    gemm(dpre, ctx.w0, acc, R, K, N, (0, R * N, N, 1), (0, wstride, 1, N), (0, 0, K), dt, nb=(1, Z), c_f32=1, accumulate=1)
                else:
                    dx = torch.empty_like(x)
                    gemm(dpre, ctx.w0, dx, R, K, N, (0, R * N, N, 1), (0, wstride, 1, N), (0, R * K, K), dt, nb=(1, Z))
            # dW_z += x_z^T dpre_z ; db_z += column sums (fused)
            with wgrad_stream(1, x, dpre):
                queued = gemm(x, dpre, ctx.gw0, K, N, R, (0, 0 if shared_x else R * K, 1, K), (0, R * N, N, 1), (0, gwstride, N), dt,
                              nb=(1, Z), c_f32=1, accumulate=1, splitk=0, colsum=ctx.gb0, sBias=(0, bstride), post=ctx.fold)
        if acc is not None:
            dx = acc.to(x.dtype).view(xshape)
        if ctx.fold is not None and not queued:
            with wgrad_stream(1, x, dpre):
                ctx.fold()
        return (dx,) + (None,) * 13


def linear_z(x, trig, w0, wstride, b0, bstride, gw0, gwstride, gb0, Z, act=ACT_NONE, shared_x=False, fold=None, grad_is_pre=False):
    """grad_is_pre (act = ELU): contract with the single consumer of the output -- it returns the gradient already multiplied by ELU'(y)
    (upconv_add(skips_pre=True)), so the separate ELU' pass is skipped."""
    return _LinearZ.apply(x, trig, w0, wstride, b0, bstride, gw0, gwstride, gb0, Z, act, shared_x, fold, grad_is_pre)


# ----------------------------------------------------------------------------------------------------
# unary activations
# ----------------------------------------------------------------------------------------------------
class _Unary(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op, p0):
        _req_cuda(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        call('stj_unary_fwd', _p(x), _p(y), x.numel(), op, float(p0), _dt(x), _st())
        ctx.op, ctx.p0 = op, p0
        ctx.save_for_backward(x if op == U_GELU else y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (s,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        call('stj_unary_bwd', _p(dy), _p(s), _p(dx), dy.numel(), ctx.op, float(ctx.p0), _dt(dy), _st())
        return dx, None, None


def gelu(x):
    return _Unary.apply(x, U_GELU, 0.0)


def elu(x):
    return _Unary.apply(x, U_ELU, 0.0)


def tanh_scale(x, s):
    return _Unary.apply(x, U_TANHS, s)


# ----------------------------------------------------------------------------------------------------
# LayerNorm (optionally fused with the PatchMerging 2x2 gather)
# ----------------------------------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g_master, b_master, pg, pb, eps, gather_res, group_rows, ngroups, gstride, with_skip, res=None):
        _req_cuda(x)
        x = x.contiguous()
        dt = _dt(x)
        if gather_res:
            B, L, C0 = x.shape
            assert L == gather_res * gather_res
            C = 4 * C0
            rows = B * (gather_res // 2) ** 2
            oshape = (B, (gather_res // 2) ** 2, C)
        else:
            C0 = 0
            C = x.shape[-1]
            rows = x.numel() // C
            oshape = x.shape
        y = torch.empty(oshape, dtype=x.dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        if res is not None:     # y = LN(x) + res in the same pass
            if gather_res or res.numel() != rows * C:
                raise RuntimeError('layernorm: res needs the plain (no gather) form and the output shape')
            call('stj_layernorm_res_fwd', _p(x), _p(pg.master), _p(pb.master), _p(res.contiguous()), _p(y), _p(mean), _p(rstd), rows, C,
                 float(eps), group_rows, ngroups, gstride, dt, _st())
        else:
            call('stj_layernorm_fwd', _p(x), _p(pg.master), _p(pb.master), _p(y), _p(mean), _p(rstd), rows, C, float(eps),
                 gather_res, C0, group_rows, ngroups, gstride, dt, _st())
        ctx.has_res = res is not None
        ctx.res_shape = res.shape if res is not None else None
        ctx.pg, ctx.pb, ctx.geo = pg, pb, (rows, C, gather_res, C0, group_rows, ngroups, gstride)
        ctx.save_for_backward(x, mean, rstd)
        if with_skip:           # second output: x itself, for the residual connection that bypasses the norm
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, mean, rstd = ctx.saved_tensors
        rows, C, gres, C0, group_rows, ngroups, gstride = ctx.geo
        if dy is None:          # only the skip output was used
            return (dskip,) + (None,) * 11
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dres = dskip.contiguous() if dskip is not None else None
        pg, pb = ctx.pg, ctx.pb
        if ngroups <= 1 and pg.part is not None and pb.part is not None:
            dg, db, nparts, pstride = pg.part[0], pb.part[0], pg.part[1], pg.part[2]
        else:
            dg, db, nparts, pstride = pg.grad, pb.grad, 1, 0
        call('stj_layernorm_bwd', _p(dy), _p(x), _p(pg.master), _p(mean), _p(rstd), _p(dx), _p(dg),
             _p(db), rows, C, gres, C0, group_rows, ngroups, gstride, _p(dres), nparts, pstride, _dt(x), _st())
        return (dx,) + (None,) * 10 + ((dy.view(ctx.res_shape) if ctx.has_res else None),)


def layernorm(x, pg, pb, eps, gather_res=0, group_rows=0, ngroups=1, gstride=0, res=None):
    """pg/pb: Param of gamma/beta (of group 0 when ngroups > 1; group g's live gstride*g elements further).
    res (same shape as the output): returns LayerNorm(x) + res from the one pass."""
    return _LayerNorm.apply(x, pg.master, pb.master, pg, pb, eps, gather_res, group_rows, ngroups, gstride, False, res)


def layernorm_skip(x, pg, pb, eps):
    """-> (LayerNorm(x), x).  Use the second output for the residual branch `x + f(LayerNorm(x))`: both gradients then arrive
    in ONE backward call and the kernel adds the skip gradient while writing dx (no separate accumulation pass)."""
    return _LayerNorm.apply(x, pg.master, pb.master, pg, pb, eps, 0, 0, 1, 0, True)


def _swin_ws(x, M, C):
    """f32 workspace of the fused Swin kernels where they split a row block / window over several workgroups (C = 384; C = 192 below
    32768 rows): partial sums of the hidden / head slices (stj_swin_split_workspace_bytes; 0 bytes = none needed)."""
    from ._lib import lib
    nbytes = int(lib().stj_swin_split_workspace_bytes(M, C))
    if nbytes == 0:
        return None
    # ONE buffer per (device, size, model), reused by every block of that width in stream order (the kernel and its finishing launch are
    # through with it before the next block's kernel starts: a model runs its encoder stages on one stream): allocating it per call grew the
    # graph's private pool by 25-100 MB per block.  Keyed by the model's arena, NOT by the stream: the capture stream of a hipGraph is not
    # the stream of the warm-up steps, and a first use inside the capture put the zero fill (49 + 9 us, in front of stages 1 and 2) into
    # every replay (profiles/r05_b_timeline_concurrent.txt)
    key = (str(x.device), int(M), int(C))
    ws = _ARENA.split_ws.get(key)        # (owned by the arena = by the model: freed with it, never handed to another model)
    if ws is None:
        # zeroed ONCE: it starts with the arrival counters of the kernels whose slices meet inside the launch (they re-arm themselves)
        ws = _ARENA.split_ws[key] = torch.zeros(nbytes // 4, dtype=torch.float32, device=x.device)
    # Two launches of one (M, C) on DIFFERENT streams would share slabs and tickets: a stream that takes the buffer over is ordered behind
    # everything the previous user has been given.  (STrajNet never needs it -- its concurrent branches are C = 96, which takes no
    # workspace -- but direct users of ops.* on two streams do.  Not across a capture boundary: the warm-up stream is synchronised before
    # the capture starts, and a capturing stream cannot wait for a stream outside its graph.)
    cur = torch.cuda.current_stream(x.device)
    last = _ARENA.split_ws_stream.get(key)
    if last is not None and last != cur and not torch.cuda.is_current_stream_capturing() and not _ARENA.split_ws_captured.get(key, False):
        cur.wait_stream(last)
    _ARENA.split_ws_stream[key] = cur
    _ARENA.split_ws_captured[key] = torch.cuda.is_current_stream_capturing()
    return ws


# ----------------------------------------------------------------------------------------------------
# fused MLP half of a Swin block: x + DropPath(fc2(gelu(fc1(LayerNorm(x)))))  -- one kernel per direction (csrc/swin_fused.hip)
# ----------------------------------------------------------------------------------------------------
class _SwinMlp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, trig, pg, pb, pw1, pb1, pw2, pb2, eps, drop, rows_per_sample):
        _req_cuda(x)
        x = x.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        p_drop, state, site = drop if drop is not None else (0.0, None, 0)
        call('stj_swin_mlp_fwd', _p(x), _p(pg.master), _p(pb.master), _p(pw1.c), _p(pb1.master), _p(pw2.c), _p(pb2.master), _p(y),
             M, C, float(eps), _p(state), site, float(p_drop), rows_per_sample, _dt(x), _p(_swin_ws(x, M, C)), _st())
        ctx.ps = (pg, pb, pw1, pb1, pw2, pb2)
        ctx.args = (M, C, float(eps), drop, rows_per_sample)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        pg, pb, pw1, pb1, pw2, pb2 = ctx.ps
        M, C, eps, drop, rps = ctx.args
        dt = _dt(x)
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        h = torch.empty((M, 4 * C), dtype=x.dtype, device=x.device)
        dpre = torch.empty_like(h)
        ln = torch.empty((M, C), dtype=x.dtype, device=x.device)
        p_drop, state, site = drop if drop is not None else (0.0, None, 0)
        dys = torch.empty_like(ln) if drop is not None else None
        if pg.part is not None and pb.part is not None:
            dg, db, nparts, pstride = pg.part[0], pb.part[0], pg.part[1], pg.part[2]
        else:
            dg, db, nparts, pstride = pg.grad, pb.grad, 1, 0
        call('stj_swin_mlp_bwd', _p(x), _p(dy), _p(pg.master), _p(pb.master), _p(pw1.c), _p(pb1.master), _p(pw2.c), _p(dx), _p(h),
             _p(dpre), _p(ln), _p(dys), _p(dg), _p(db), nparts, pstride, M, C, eps, _p(state), site, float(p_drop), rps, dt, _p(_swin_ws(x, M, C)), _st())
        g2 = dys if dys is not None else dy.view(M, C)
        with wgrad_stream(1, ln, dpre, h, g2), gemm_group(M < _GROUP_MAX_ROWS):
            gemm(ln, dpre, pw1.grad, C, 4 * C, M, (0, 0, 1, C), (0, 0, 4 * C, 1), (0, 0, 4 * C), dt, c_f32=1, accumulate=1,
                 splitk=0, colsum=pb1.grad)                               # dW1 += LN(x)^T dpre ; db1 += 1^T dpre
            gemm(h, g2, pw2.grad, 4 * C, C, M, (0, 0, 1, 4 * C), (0, 0, C, 1), (0, 0, C), dt, c_f32=1, accumulate=1,
                 splitk=0, colsum=pb2.grad)                               # dW2 += gelu(pre)^T (dp dy) ; db2 += 1^T (dp dy)
        return (dx,) + (None,) * 10


def swin_mlp(x, pg, pb, pw1, pb1, pw2, pb2, eps, dctx=None, name=None, p_drop=0.0, rows_per_sample=None):
    """x [..., C] -> x + DropPath(Mlp(LayerNorm(x)))  (modules.py:260).  dctx/name/p_drop: the DropPath site of a training step (one
    draw per run of rows_per_sample rows = per sample)."""
    C = x.shape[-1]
    M = x.numel() // C
    rps = int(rows_per_sample) if rows_per_sample else M
    drop = None
    if dctx is not None and p_drop > 0.0:
        drop = (float(p_drop), dctx.snap, dctx.site(name, (M // rps,), p_drop))
    return _SwinMlp.apply(x, pg.master, pg, pb, pw1, pb1, pw2, pb2, eps, drop, rps)


def _trig(t):
    """The 'some parameter requires grad' trigger argument of the fused ops.  ctx.needs_input_grad stays True under torch.no_grad()
    for a tensor that requires grad (and grad mode reads as off inside every forward()), so with grad mode off the ops get a detached
    alias: `train` is then False and nothing is saved for a backward pass that cannot happen."""
    return t if torch.is_grad_enabled() else t.detach()


class _SwinAttnHalf(torch.autograd.Function):
    """x -> x + DropPath(proj(W-MSA / SW-MSA(LN(x)))): forward is ONE kernel (stj_swin_attn_fwd); backward runs the proj / qkv
    input and weight gradients as GEMMs around the window-attention backward kernel, on the operands the forward kernel saved."""
    @staticmethod
    def forward(ctx, x, trig, pg, pb, pwq, pbq, pt, pwp, pbp, B, res, shift, eps, drop):
        _req_cuda(x)
        x = x.contiguous()
        C = x.shape[-1]
        N = res * res
        y = torch.empty_like(x)
        train = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])     # False under torch.no_grad() (see _trig): nothing is saved
        qkv = a = ln = mean = rstd = None
        if train:
            qkv = torch.empty((B, N, 3 * C), dtype=x.dtype, device=x.device)
            a = torch.empty((B, N, C), dtype=x.dtype, device=x.device)
            ln = torch.empty((B, N, C), dtype=x.dtype, device=x.device)
            mean = torch.empty(B * N, dtype=torch.float32, device=x.device)
            rstd = torch.empty(B * N, dtype=torch.float32, device=x.device)
        p_drop, state, site = drop if drop is not None else (0.0, None, 0)
        call('stj_swin_attn_fwd', _p(x), _p(pg.master), _p(pb.master), _p(pwq.c), _p(pbq.master), _p(pt.master), _p(pwp.c),
             _p(pbp.master), _p(y), _p(qkv), _p(a), _p(ln), _p(mean), _p(rstd), B, res, C, shift, float(eps), _p(state), site,
             float(p_drop), _dt(x), _p(_swin_ws(x, B * N, C)), _st())
        ctx.ps = (pg, pb, pwq, pbq, pt, pwp, pbp)
        ctx.args = (B, res, C, shift, drop)
        ctx.save_for_backward(x, qkv, a, ln, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, qkv, a, ln, mean, rstd = ctx.saved_tensors
        pg, pb, pwq, pbq, pt, pwp, pbp = ctx.ps
        B, res, C, shift, drop = ctx.args
        N = res * res
        M = B * N
        heads = C // 32
        dt = _dt(x)
        dy = dy.contiguous()
        items = B * (res // 8) ** 2 * heads
        tparts = 32 if items >= 1024 else (16 if items >= 256 else 1)
        if pg.part is not None and pb.part is not None:
            dg, db, np_, ps_ = pg.part[0], pb.part[0], pg.part[1], pg.part[2]
        else:
            dg, db, np_, ps_ = pg.grad, pb.grad, 1, 0
        dx = torch.empty_like(x)
        dqkv = torch.empty_like(qkv)
        a2, ln2, dq2 = a.view(M, C), ln.view(M, C), dqkv.view(M, 3 * C)
        if FUSED_ATTN_BWD:
            # ONE kernel per window: DropPath factor, proj dgrad, window-attention backward, qkv dgrad, LN backward + shortcut
            p_drop, state, site = drop if drop is not None else (0.0, None, 0)
            dys = torch.empty_like(dy) if drop is not None else None
            if pt.part is not None:
                dtab, tp, own = pt.part[0], min(tparts, pt.part[1]), None
            else:
                own = zeros_f32((tparts,) + tuple(pt.grad.shape), x.device)
                dtab, tp = own, tparts
            call('stj_swin_attn_bwd', _p(x), _p(dy), _p(qkv), _p(mean), _p(rstd), _p(pg.master), _p(pwq.c), _p(pwp.c), _p(pt.master),
                 _p(dx), _p(dqkv), _p(dys), _p(dtab), tp, _p(dg), _p(db), np_, ps_, B, res, C, shift, _p(state), site, float(p_drop),
                 dt, _p(_swin_ws(x, M, C)), _st())
            if own is not None:
                pt.grad.add_(own.sum(0))
            dys2 = (dys if dys is not None else dy).view(M, C)
        else:
            dys = dy
            if drop is not None:            # gradient of the branch = DropPath factor * dy (same draw, re-derived)
                p_drop, state, site = drop
                dys = torch.empty_like(dy)
                call('stj_dropout', _p(dy), None, _p(dys), dy.numel(), N * C, float(p_drop), _p(state), site, dt, _st())
            dys2 = dys.view(M, C)
            da = torch.empty_like(a)
            gemm(dys2, pwp.c, da, M, C, C, (0, 0, C, 1), (0, 0, 1, C), (0, 0, C), dt)                    # da = dys Wp^T
            if pt.part is not None:
                call('stj_win_attn_bwd', _p(qkv), _p(pt.master), _p(da), _p(dqkv), _p(pt.part[0]), min(tparts, pt.part[1]), B, res, heads,
                     shift, dt, _st())
            else:
                part = zeros_f32((tparts,) + tuple(pt.grad.shape), x.device)
                call('stj_win_attn_bwd', _p(qkv), _p(pt.master), _p(da), _p(dqkv), _p(part), tparts, B, res, heads, shift, dt, _st())
                pt.grad.add_(part.sum(0))
            dln = torch.empty_like(ln)
            gemm(dq2, pwq.c, dln, M, C, 3 * C, (0, 0, 3 * C, 1), (0, 0, 1, 3 * C), (0, 0, C), dt)         # dln = dqkv Wqkv^T
            call('stj_layernorm_bwd', _p(dln), _p(x), _p(pg.master), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), M, C, 0, 0, 0, 1, 0,
                 _p(dy), np_, ps_, dt, _st())                                                             # + the shortcut gradient
        with wgrad_stream(1, a2, dys2, ln2, dq2), gemm_group(M < _GROUP_MAX_ROWS):
            gemm(a2, dys2, pwp.grad, C, C, M, (0, 0, 1, C), (0, 0, C, 1), (0, 0, C), dt, c_f32=1, accumulate=1, splitk=0,
                 colsum=pbp.grad)                                                                         # dWp += a^T dys ; dbp
            gemm(ln2, dq2, pwq.grad, C, 3 * C, M, (0, 0, 1, C), (0, 0, 3 * C, 1), (0, 0, 3 * C), dt, c_f32=1, accumulate=1,
                 splitk=0, colsum=pbq.grad)                                                               # dWqkv += ln^T dqkv ; dbqkv
        return (dx,) + (None,) * 13


FUSED_ATTN_BWD = os.environ.get('STJ_FUSED_SWIN', '1') != '0'       # (STJ_FUSED_SWIN=0: the layer-by-layer Swin block, the form the f32 mode's C = 384 stage takes)


def swin_attn_half(x, pg, pb, pwq, pbq, pt, pwp, pbp, B, res, shift, eps, dctx=None, name=None, p_drop=0.0):
    """x [B, res*res, C] -> x + DropPath(proj(window_attention(LN(x))))  (modules.py:225-258)."""
    drop = None
    if dctx is not None and p_drop > 0.0:
        drop = (float(p_drop), dctx.snap, dctx.site(name, (B,), p_drop))
    return _SwinAttnHalf.apply(x, _trig(pg.master), pg, pb, pwq, pbq, pt, pwp, pbp, B, res, shift, eps, drop)


def _attn_cost(a):
    B, res, C, dt = a[14], a[15], a[16], a[22]
    es = 4 if dt == 0 else 2
    M = B * res * res
    fl = 2.0 * M * C * 4 * C + M * 4.0 * 64 * C
    tens = 2 + (5 if getattr(a[9], 'value', None) else 0)
    return f'swin_attn_fwd[B{B} {res}x{res} C{C}]', 'swin_attn_fwd', fl, fl, es * M * C * tens


prof.EXTRA_MODELS['stj_swin_attn_fwd'] = _attn_cost


def _attn_bwd_cost(a):
    B, res, C, dt = a[18], a[19], a[20], a[25]
    es = 4 if dt == 0 else 2
    M = B * res * res
    fl = 2.0 * M * C * 4 * C + M * 10.0 * 64 * C
    return f'swin_attn_bwd[B{B} {res}x{res} C{C}]', 'swin_attn_bwd', fl, fl, es * M * C * (3 + 3 + 3 + (1 if getattr(a[11], 'value', None) else 0))


prof.EXTRA_MODELS['stj_swin_attn_bwd'] = _attn_bwd_cost


def _mlp_cost(kind):
    def f(a):
        M, C, dt = (a[8], a[9], a[15]) if kind == 'fwd' else (a[16], a[17], a[23])
        es = 4 if dt == 0 else 2
        fl = 2.0 * M * C * 4 * C * (2 if kind == 'fwd' else 4)
        by = es * M * C * 2 if kind == 'fwd' else es * M * (C * 5 + 8 * C)
        return f'swin_mlp_{kind}[{M}x{C}]', 'swin_mlp_' + kind, fl, fl, by
    return f


prof.EXTRA_MODELS['stj_swin_mlp_fwd'] = _mlp_cost('fwd')
prof.EXTRA_MODELS['stj_swin_mlp_bwd'] = _mlp_cost('bwd')


# ----------------------------------------------------------------------------------------------------
# fused (shifted) window attention
# ----------------------------------------------------------------------------------------------------
class _WinAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, t_master, pt, B, res, heads, shift):
        _req_cuda(qkv)
        qkv = qkv.contiguous()
        C = heads * 32
        out = torch.empty((B, res * res, C), dtype=qkv.dtype, device=qkv.device)
        call('stj_win_attn_fwd', _p(qkv), _p(pt.master), _p(out), B, res, heads, shift, _dt(qkv), _st())
        ctx.pt, ctx.geo = pt, (B, res, heads, shift)
        ctx.save_for_backward(qkv)
        return out

    @staticmethod
    def backward(ctx, dout):
        (qkv,) = ctx.saved_tensors
        B, res, heads, shift = ctx.geo
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        items = B * (res // 8) ** 2 * heads
        pt = ctx.pt
        nparts = 32 if items >= 1024 else (16 if items >= 256 else 1)
        if pt.part is not None:     # model-owned copies, folded into .grad once per step
            call('stj_win_attn_bwd', _p(qkv), _p(pt.master), _p(dout), _p(dqkv), _p(pt.part[0]), min(nparts, pt.part[1]), B, res, heads,
                 shift, _dt(qkv), _st())
        elif nparts == 1:
            call('stj_win_attn_bwd', _p(qkv), _p(pt.master), _p(dout), _p(dqkv), _p(pt.grad), 1, B, res, heads, shift,
                 _dt(qkv), _st())
        else:       # the workgroups spread their bias-table atomics over nparts copies (same-address contention), summed here
            part = torch.zeros((nparts,) + tuple(pt.grad.shape), dtype=torch.float32, device=qkv.device)
            call('stj_win_attn_bwd', _p(qkv), _p(pt.master), _p(dout), _p(dqkv), _p(part), nparts, B, res, heads, shift,
                 _dt(qkv), _st())
            pt.grad.add_(part.sum(0))
        return dqkv, None, None, None, None, None, None


def win_attn(qkv, pt, B, res, heads, shift):
    return _WinAttn.apply(qkv, pt.master, pt, B, res, heads, shift)


# ----------------------------------------------------------------------------------------------------
# global multi-head attention core: softmax(scale * q k^T (+bias) (+mask)) v   (projections are `linear`s)
# ----------------------------------------------------------------------------------------------------
class _MhaCore(torch.autograd.Function):
    """q [Bt,Nq,H*d], k,v [Bt,Nk,H*d]; qvalid [Bt,Nq] / kvalid [Bt,Nk] int32 or None; bias f32 [Bt,H,Nq,Nk] or None.
    Returns o [Bt,Nq,H*d].  Gradient w.r.t. bias is returned in the activation dtype.
    fg = (table Param, Hh, Ww) with `off` [Bt,H,Nq,2]: the FG-MSA sampled relative-position bias is built here from the offsets
    (stj_fg_bias_fwd) and its backward consumes dS directly -- as a separate autograd node the [Bt,H,Nq,Nk] gradient crossed the
    f32 <-> activation dtype boundary twice (two cast kernels over 17 MB at B=8)."""
    @staticmethod
    def forward(ctx, q, k, v, bias, H, d, scale, qvalid, kvalid, drop, off=None, t_master=None, fg=None):
        _req_cuda(q, k, v)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        Bt, Nq, HD = q.shape
        Nk = k.shape[1]
        dt = _dt(q)
        S = torch.empty((Bt, H, Nq, Nk), dtype=torch.float32, device=q.device)
        # S[b,h] = scale * Q[b,:,h,:] K[b,:,h,:]^T
        gemm(q, k, S, Nq, Nk, d, (Nq * HD, d, HD, 1), (Nk * HD, d, 1, HD), (H * Nq * Nk, Nq * Nk, Nk), dt, nb=(Bt, H),
             alpha=scale, c_f32=1)
        P = torch.empty((Bt, H, Nq, Nk), dtype=q.dtype, device=q.device)
        if fg is not None:
            off = off.contiguous()
            bias = torch.empty((Bt, H, Nq, Nk), dtype=torch.float32, device=q.device)
            call('stj_fg_bias_fwd', _p(off), _p(fg[0].master), _p(bias), Bt, H, fg[1], fg[2], dt, _st())
        elif bias is not None:
            bias = bias.contiguous()
        call('stj_softmax_fwd', _p(S), _p(P), _p(qvalid), _p(kvalid), _p(bias), Bt, H, Nq, Nk, dt, _st())
        Pd = P
        if drop is not None:            # tfa-MHA: dropout on the attention coefficients, after the softmax (App. C-1)
            p_drop, state, site = drop
            Pd = torch.empty_like(P)
            call('stj_dropout', _p(P), None, _p(Pd), P.numel(), 1, float(p_drop), _p(state), site, dt, _st())
        o = torch.empty_like(q)
        # O[b,:,h,:] = P[b,h] V[b,:,h,:]
        gemm(Pd, v, o, Nq, d, Nk, (H * Nq * Nk, Nq * Nk, Nk, 1), (Nk * HD, d, HD, 1), (Nq * HD, d, HD), dt, nb=(Bt, H))
        ctx.geo = (Bt, Nq, Nk, H, d, scale, bias is not None and fg is None)
        ctx.drop, ctx.fg = drop, fg
        ctx.save_for_backward(q, k, v, P, Pd if drop is not None else None, off if fg is not None else None)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, P, Pd, off = ctx.saved_tensors
        if Pd is None:
            Pd = P
        Bt, Nq, Nk, H, d, scale, has_bias = ctx.geo
        HD = H * d
        dt = _dt(q)
        do = do.contiguous()
        dP = torch.empty((Bt, H, Nq, Nk), dtype=torch.float32, device=q.device)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        with gemm_group():
            # dP[b,h] = dO[b,:,h,:] V[b,:,h,:]^T ; dV[b,:,h,:] = P[b,h]^T dO
            gemm(do, v, dP, Nq, Nk, d, (Nq * HD, d, HD, 1), (Nk * HD, d, 1, HD), (H * Nq * Nk, Nq * Nk, Nk), dt, nb=(Bt, H), c_f32=1)
            gemm(Pd, do, dv, Nk, d, Nq, (H * Nq * Nk, Nq * Nk, 1, Nk), (Nq * HD, d, HD, 1), (Nk * HD, d, HD), dt, nb=(Bt, H))
        if ctx.drop is not None:        # same mask, re-derived from (state, site); dP is f32
            p_drop, state, site = ctx.drop
            call('stj_dropout', _p(dP), None, _p(dP), dP.numel(), 1, float(p_drop), _p(state), site, 0, _st())
        dS = torch.empty_like(P)
        call('stj_softmax_bwd', _p(P), _p(dP), _p(dS), Bt * H * Nq, Nk, dt, _st())
        with gemm_group():        # dQ = scale dS K ; dK = scale dS^T Q
            gemm(dS, k, dq, Nq, d, Nk, (H * Nq * Nk, Nq * Nk, Nk, 1), (Nk * HD, d, HD, 1), (Nq * HD, d, HD), dt, nb=(Bt, H), alpha=scale)
            gemm(dS, q, dk, Nk, d, Nq, (H * Nq * Nk, Nq * Nk, 1, Nk), (Nq * HD, d, HD, 1), (Nk * HD, d, HD), dt, nb=(Bt, H), alpha=scale)
        doff = None
        if ctx.fg is not None:
            pt, Hh, Ww = ctx.fg
            doff32 = zeros_f32(tuple(off.shape), off.device)     # accumulated by the kernel's query slices
            call('stj_fg_bias_bwd', _p(off), _p(pt.master), _p(dS), _p(pt.grad), _p(doff32), Bt, H, Hh, Ww, dt, _st())
            doff = doff32.to(off.dtype)
        return dq, dk, dv, (dS if has_bias else None), None, None, None, None, None, None, doff, None, None


UPWG_BUDGET = 128       # workgroups of the two large up-conv weight-gradient launches inside the concurrent step (half the CUs; 256 when alone)


class _SmallAttn(torch.autograd.Function):
    """mha_core for at most 16 queries / keys per (batch element, head): one launch per direction (stj_small_attn_*)."""
    @staticmethod
    def forward(ctx, q, k, v, H, d, scale, qvalid, kvalid, drop):
        _req_cuda(q, k, v)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        Bt, N, HD = q.shape
        o = torch.empty_like(q)
        p_drop, state, site = drop if drop is not None else (0.0, None, 0)
        call('stj_small_attn_fwd', _p(q), _p(k), _p(v), _p(qvalid), _p(kvalid), _p(o), Bt, N, H, d, float(scale), _p(state), site, float(p_drop),
             _dt(q), _st())
        ctx.geo, ctx.drop = (Bt, N, H, d, float(scale)), drop
        ctx.save_for_backward(q, k, v, qvalid, kvalid)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, qvalid, kvalid = ctx.saved_tensors
        Bt, N, H, d, scale = ctx.geo
        p_drop, state, site = ctx.drop if ctx.drop is not None else (0.0, None, 0)
        do = do.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        call('stj_small_attn_bwd', _p(q), _p(k), _p(v), _p(qvalid), _p(kvalid), _p(do), _p(dq), _p(dk), _p(dv), Bt, N, H, d, scale, _p(state),
             site, float(p_drop), _dt(q), _st())
        return dq, dk, dv, None, None, None, None, None, None


SMALL_ATTN = True       # False: the layer-by-layer path for every geometry (tests compare the two)


def mha_core(q, k, v, H, d, scale, qvalid=None, kvalid=None, bias=None, drop=None, fg_off=None, fg=None):
    """drop = (p, state, site) applies attention dropout to the softmax output (training).
    fg_off [Bt,H,Nq,2] + fg = (table Param, Hh, Ww): FG-MSA bias sampled from the offsets inside the op."""
    if SMALL_ATTN and fg is None and bias is None and q.shape[1] == k.shape[1]:
        from ._lib import lib
        if lib().stj_small_attn_supported(q.shape[1], H, d, _dt(q)):
            return _SmallAttn.apply(q, k, v, H, d, scale, qvalid, kvalid, drop)
    if fg is not None:
        return _MhaCore.apply(q, k, v, None, H, d, scale, qvalid, kvalid, drop, fg_off, fg[0].master, fg)
    return _MhaCore.apply(q, k, v, bias, H, d, scale, qvalid, kvalid, drop)


# ----------------------------------------------------------------------------------------------------
# fused FG-MSA attention core (csrc/fgattn.hip): one kernel per direction
# ----------------------------------------------------------------------------------------------------
class _FgAttn(torch.autograd.Function):
    """a = softmax(scale q k^T + bias(off, table)) v (FG_MSA.py:150-176) for q, k, v [B,HW,G*48], off [B,G,HW,2], table Param
    [2Hh-1,2Ww-1,G]; the logits / bias / probabilities stay on chip.  Same arguments and results as mha_core(..., fg_off=, fg=)."""
    @staticmethod
    def forward(ctx, q, k, v, off, t_master, pt, Hh, Ww, scale):
        _req_cuda(q, k, v, off)
        q, k, v, off = q.contiguous(), k.contiguous(), v.contiguous(), off.contiguous()
        B, HW, C = q.shape
        G = off.shape[1]
        a = torch.empty_like(q)
        train = any(ctx.needs_input_grad[:5])
        lse = torch.empty((B, G, HW), dtype=torch.float32, device=q.device) if train else None
        call('stj_fg_attn_fwd', _p(q), _p(k), _p(v), _p(off), _p(pt.master), _p(a), _p(lse), B, G, Hh, Ww, float(scale), _dt(q), _st())
        ctx.geo = (B, G, Hh, Ww, float(scale))
        ctx.pt = pt
        ctx.save_for_backward(q, k, v, off, a, lse)
        return a

    @staticmethod
    def backward(ctx, da):
        from ._lib import lib
        q, k, v, off, a, lse = ctx.saved_tensors
        B, G, Hh, Ww, scale = ctx.geo
        pt = ctx.pt
        da = da.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        n = int(lib().stj_fg_attn_bwd_workspace_bytes(B, G, Hh, Ww)) // 4
        part = torch.empty((2, n), dtype=torch.float32, device=q.device)
        doff32 = zeros_f32(tuple(off.shape), off.device)          # accumulated by the kernel's query tiles
        call('stj_fg_attn_bwd', _p(q), _p(k), _p(v), _p(off), _p(pt.master), _p(a), _p(lse), _p(da), _p(dq), _p(dk), _p(dv), _p(part[0]),
             _p(part[1]), _p(pt.grad), _p(doff32), B, G, Hh, Ww, scale, _dt(q), _st())
        return dq, dk, dv, doff32.to(off.dtype), None, None, None, None, None


def fg_attn_ok(dtype, Hh, Ww, gc):
    """The geometries the fused FG-MSA kernel covers: 48-wide groups, an 8 x 8 or 16 x 16 map (every storage type: the f32 parity mode
    runs the same kernel template with exact-f32 MFMA)."""
    return gc == 48 and Hh == Ww and Hh in (8, 16)


def fg_attn(q, k, v, off, pt, Hh, Ww, scale):
    return _FgAttn.apply(q, k, v, off, _trig(pt.master), pt, Hh, Ww, scale)


def _fgattn_cost(kind):
    def f(a):
        if kind == 'fwd':
            B, G, Hh, Ww, train = a[7], a[8], a[9], a[10], bool(getattr(a[6], 'value', None))
        else:
            B, G, Hh, Ww, train = a[15], a[16], a[17], a[18], True
        HW = Hh * Ww
        fl = 2.0 * B * G * HW * HW * 48 * (2 if kind == 'fwd' else 5)
        act = 2 * B * HW * G * 48
        if kind == 'fwd':
            by = 4 * act + (4 * B * G * HW if train else 0)
        else:
            by = 8 * act + 2 * (HW // 64) * 2 * act * 2           # + the f32 per-tile partials, written and read back
        by += B * G * HW * 2 * 2
        return f'fgattn_{kind}[B{B} G{G} {Hh}x{Ww}]', 'fgattn_' + kind, fl, fl, by
    return f


prof.EXTRA_MODELS['stj_fg_attn_fwd'] = _fgattn_cost('fwd')
prof.EXTRA_MODELS['stj_fg_attn_bwd'] = _fgattn_cost('bwd')


# ----------------------------------------------------------------------------------------------------
# fused Cross_AttentionT block x Z weight sets (csrc/xattn_fused.hip): one kernel per direction
# ----------------------------------------------------------------------------------------------------
def xattn_pack(ps, zstride, Z, dtype, out=None):
    """The Z weight sets' packed LDS-image stream (stj_xattn_pack) in `dtype`; ps = set-0 Params (see xattn).  Once per step."""
    from ._lib import lib
    dev = ps['wq'].master.device
    nbytes = int(lib().stj_xattn_pack_workspace_bytes(DTYPE_CODE[dtype]))
    if out is None:         # (+ the tail the kernels' fixed-size chunk copies may read past the last set)
        out = torch.zeros(Z * nbytes + int(lib().stj_xattn_pack_tail_workspace_bytes(DTYPE_CODE[dtype])), dtype=torch.uint8, device=dev)
    call('stj_xattn_pack', _p(ps['wq'].master), _p(ps['wo'].master), _p(ps['w1'].master), _p(ps['w2'].master), zstride, Z, _p(out),
         DTYPE_CODE[dtype], _st())
    return out


XATTN_WG_KIND = 1      # (2 = on the up-conv weight-gradient side stream: 1250 vs 1335 scenes/s, round 5 -- one more fork off the main chain and the graph executor serialises the deferred up-conv weight gradients with it)


class _XAttn(torch.autograd.Function):
    """y = LN2(FFN(LN1(MHA(query, k, v)))) + query for Z weight sets (trajNet.py:224-234,305-317).  query [Z,B,HW,384]; k, v [Z,B*64,126]
    (projected keys / values: their projections stay autograd nodes of their own); ps: Params of set 0 (set z lies zstride elements
    further in the flat buffers).  Backward = ONE kernel + the dk / dv tile reduction + one grouped launch of the four weight-gradient
    GEMMs, on operands the backward kernel writes once."""
    @staticmethod
    def forward(ctx, query, k, v, trig, kvalid, pack, ps, zstride, drop, defer_wg=False):
        _req_cuda(query, k, v)
        query, k, v = query.contiguous(), k.contiguous(), v.contiguous()
        Z, B, HW, Cb = query.shape
        dt = _dt(query)
        y = torch.empty_like(query)
        train = any(ctx.needs_input_grad[:4])
        sq = so = sv1 = su2 = None
        if train:
            sq = torch.empty((Z, B, HW, 144), dtype=query.dtype, device=query.device)
            so = torch.empty_like(sq)
            sv1 = torch.empty((Z, B, HW, 128), dtype=query.dtype, device=query.device)
            su2 = torch.empty_like(query)
        p_drop, state, sites = drop if drop is not None else (0.0, None, (0, 0, 0))
        call('stj_xattn_fwd', _p(query), _p(k), _p(v), _p(kvalid), _p(pack), _p(ps['bo'].master), _p(ps['g1'].master), _p(ps['be1'].master),
             _p(ps['b1'].master), _p(ps['b2'].master), _p(ps['g2'].master), _p(ps['be2'].master), zstride, _p(y), _p(sq), _p(so), _p(sv1),
             _p(su2), Z, B, HW, _p(state), sites[0], sites[1], sites[2], float(p_drop), dt, _st())
        ctx.ps, ctx.zstride, ctx.drop, ctx.pack, ctx.defer_wg = ps, zstride, drop, pack, defer_wg
        ctx.save_for_backward(query, k, v, kvalid, sq, so, sv1, su2)
        return y

    @staticmethod
    def backward(ctx, dy):
        from ._lib import lib
        query, k, v, kvalid, sq, so, sv1, su2 = ctx.saved_tensors
        ps, zs, drop = ctx.ps, ctx.zstride, ctx.drop
        Z, B, HW, Cb = query.shape
        R = B * HW
        dt = _dt(query)
        dev, ty = query.device, query.dtype
        dy = dy.contiguous()
        dquery = torch.empty_like(query)
        dk, dv = torch.empty_like(k), torch.empty_like(v)
        wsn = int(lib().stj_xattn_bwd_workspace_bytes(Z, B, HW)) // 4
        dkp = torch.empty(wsn, dtype=torch.float32, device=dev)
        dvp = torch.empty(wsn, dtype=torch.float32, device=dev)
        hd = torch.empty((Z, R, 512), dtype=ty, device=dev)
        dpre = torch.empty_like(hd)
        du2 = torch.empty((Z, R, 384), dtype=ty, device=dev)
        n1 = torch.empty((Z, R, 128), dtype=ty, device=dev)
        dv1 = torch.empty_like(n1)
        dq = torch.empty((Z, R, 144), dtype=ty, device=dev)
        p_drop, state, sites = drop if drop is not None else (0.0, None, (0, 0, 0))
        call('stj_xattn_bwd', _p(dy), _p(query), _p(k), _p(v), _p(kvalid), _p(ctx.pack), _p(ps['g1'].master), _p(ps['be1'].master),
             _p(ps['b1'].master), _p(ps['g2'].master), zs, _p(sq), _p(sv1), _p(su2), _p(dquery), _p(dk), _p(dv), _p(dkp), _p(dvp), _p(hd),
             _p(dpre), _p(du2), _p(n1), _p(dv1), _p(dq), _p(ps['g1'].grad), _p(ps['be1'].grad), _p(ps['bo'].grad), _p(ps['g2'].grad),
             _p(ps['be2'].grad), Z, B, HW, _p(state), sites[0], sites[1], sites[2], float(p_drop), dt, _st())
        # the four weight gradients of the Z sets, straight into the flat gradient buffer: one grouped launch -- here, or (defer_wg) handed to
        # the key / value projections' backward, which runs on the agent branch's stream: nobody on the main chain waits for them
        def wg():
          with wgrad_stream(XATTN_WG_KIND, hd, dpre, du2, n1, dv1, dq, so, query), gemm_group():
            gemm(hd, du2, ps['w2'].grad, 512, 384, R, (0, R * 512, 1, 512), (0, R * 384, 384, 1), (0, zs, 384), dt, nb=(1, Z), c_f32=1,
                 accumulate=1, splitk=0, colsum=ps['b2'].grad, sBias=(0, zs))                      # dW2 += hd^T du2 ; db2
            gemm(n1, dpre, ps['w1'].grad, 128, 512, R, (0, R * 128, 1, 128), (0, R * 512, 512, 1), (0, zs, 512), dt, nb=(1, Z), c_f32=1,
                 accumulate=1, splitk=0, colsum=ps['b1'].grad, sBias=(0, zs))                      # dW1 += n1^T dpre ; db1
            gemm(so, dv1, ps['wo'].grad, 42, 128, R, (R * 144, 48, 1, 144), (R * 128, 0, 128, 1), (zs, 42 * 128, 128), dt, nb=(Z, 3),
                 c_f32=1, accumulate=1, splitk=0)                                                  # dWo[z,h] += O_h^T dv1
            gemm(query, dq, ps['wq'].grad, 384, 42, R, (R * 384, 0, 1, 384), (R * 144, 48, 144, 1), (zs, 384 * 42, 42), dt, nb=(Z, 3),
                 c_f32=1, accumulate=1, splitk=0)                                                  # dWq[z,h] += query^T dq_h
        if ctx.defer_wg:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            _PENDING_WG.setdefault(dev.index, []).append((wg, ev, (hd, dpre, du2, n1, dv1, dq, so, query)))
        else:
            wg()
        return dquery, dk, dv, None, None, None, None, None, None, None


def xattn(query, k, v, kvalid, pack, ps, zstride, dctx=None, names=None, p_drop=0.1, defer_wg=False):
    """Fused Cross_AttentionT x Z.  ps: dict of set-0 Params {wq, wo, bo, g1, be1, w1, b1, w2, b2, g2, be2}; dctx / names: the three
    dropout sites of a training step (attention coefficients, after FFN1, after FFN2), registered with the unfused draw shapes."""
    Z, B, HW, _ = query.shape
    drop = None
    if dctx is not None:
        sites = (dctx.site(names[0], (Z, B, 3, HW, 64), p_drop), dctx.site(names[1], (Z, B * HW, 512), p_drop),
                 dctx.site(names[2], (Z, B * HW, 384), p_drop))
        drop = (float(p_drop), dctx.snap, sites)
    return _XAttn.apply(query, k, v, _trig(ps['wq'].master), kvalid, pack, ps, zstride, drop, defer_wg and not _SERIAL)


def _xattn_cost(kind):
    def f(a):
        if kind == 'fwd':
            Z, B, HW, dt, train = a[18], a[19], a[20], a[26], bool(getattr(a[14], 'value', None))
        else:
            Z, B, HW, dt, train = a[30], a[31], a[32], a[38], True
        es = 4 if dt == 0 else 2
        rows = Z * B * HW
        macs = 384 * 126 + 2 * 64 * 126 + 126 * 128 + 128 * 512 + 512 * 384            # per token: q proj, q k^T + P v, out proj, FFN1, FFN2
        fl = 2.0 * rows * macs * (1 if kind == 'fwd' else 2)                          # (weight gradients are the caller's GEMMs)
        if kind == 'fwd':
            by = es * rows * (2 * 384 + ((144 + 144 + 128 + 384) if train else 0))
        else:
            by = es * rows * (3 * 384 + 144 + 128 + 384 + 2 * 512 + 384 + 2 * 128 + 144)
        by += Z * 707840 * es // 2 * max(1, B * HW // 64 // 32)                           # the set's weight stream (re-read from L2 by the tiles)
        return f'xattn_{kind}[Z{Z} B{B} HW{HW}]', 'xattn_' + kind, fl, fl, by
    return f


prof.EXTRA_MODELS['stj_xattn_fwd'] = _xattn_cost('fwd')
prof.EXTRA_MODELS['stj_xattn_bwd'] = _xattn_cost('bwd')


# ----------------------------------------------------------------------------------------------------
# Dropout / DropPath (+ fused residual)
# ----------------------------------------------------------------------------------------------------
class DropCtx:
    """Random-stream bookkeeping of one model: device state {seed, step}, a per-forward snapshot that the backward kernels
    re-derive their masks from, and the registry name -> (site id, draw shape, p) of the current forward."""

    def __init__(self, device, seed=0):
        self.state = torch.tensor([int(seed), 0], dtype=torch.int64, device=device)
        self.snap, self.n, self.sites = None, 0, {}

    def begin(self):
        self.snap = torch.empty_like(self.state)  # backward of THIS forward keeps reading this step, whatever runs in between
        call('stj_rng_advance_snap', _p(self.state), _p(self.snap), _st())
        self.n, self.sites = 0, {}

    def site(self, name, shape, p):
        sid = self.n
        self.n += 1
        self.sites[name] = (sid, tuple(shape), float(p), self.snap)
        return sid

    def mask(self, name):
        """Keep mask (uint8, draw shape) of a site of the last forward -- test hook."""
        sid, shape, p, snap = self.sites[name]
        n = 1
        for d in shape:
            n *= d
        m = torch.empty(n, dtype=torch.uint8, device=snap.device)
        call('stj_dropout_mask', _p(m), n, p, _p(snap), sid, _st())
        return m.view(shape)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, p, inner, state, site):
        _req_cuda(x)
        x = x.contiguous()
        r = res.contiguous() if res is not None else None
        y = torch.empty_like(x)
        call('stj_dropout', _p(x), _p(r), _p(y), x.numel(), inner, float(p), _p(state), site, _dt(x), _st())
        ctx.args = (float(p), inner, state, site, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, inner, state, site, has_res = ctx.args
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        call('stj_dropout', _p(dy), None, _p(dx), dy.numel(), inner, p, _p(state), site, _dt(dy), _st())
        return dx, (dy if has_res else None), None, None, None, None


def dropout(x, p, dctx, name, res=None, per_sample=False):
    """Keras Dropout(p) (per element) or DropPath(p) (per_sample: one draw per x[0] slice), optionally + res."""
    inner = x[0].numel() if per_sample else 1
    shape = (x.shape[0],) if per_sample else tuple(x.shape)
    site = dctx.site(name, shape, p)
    return _Dropout.apply(x, res, p, inner, dctx.snap, site)


class _FgBias(torch.autograd.Function):
    """off [B,G,HW,2] (activation dtype) + table param [2H-1,2W-1,G] -> f32 bias [B,G,HW,HW]."""
    @staticmethod
    def forward(ctx, off, t_master, pt, Hh, Ww):
        _req_cuda(off)
        off = off.contiguous()
        B, G = off.shape[0], off.shape[1]
        bias = torch.empty((B, G, Hh * Ww, Hh * Ww), dtype=torch.float32, device=off.device)
        call('stj_fg_bias_fwd', _p(off), _p(pt.master), _p(bias), B, G, Hh, Ww, _dt(off), _st())
        ctx.pt, ctx.geo = pt, (B, G, Hh, Ww)
        ctx.save_for_backward(off)
        return bias

    @staticmethod
    def backward(ctx, dbias):
        (off,) = ctx.saved_tensors
        B, G, Hh, Ww = ctx.geo
        dbias = dbias.contiguous().to(off.dtype)
        doff = zeros_f32(tuple(off.shape), off.device)     # accumulated by the kernel's query slices
        call('stj_fg_bias_bwd', _p(off), _p(ctx.pt.master), _p(dbias), _p(ctx.pt.grad), _p(doff), B, G, Hh, Ww, _dt(off), _st())
        return doff.to(off.dtype), None, None, None, None


def fg_bias(off, pt, Hh, Ww):
    return _FgBias.apply(off, pt.master, pt, Hh, Ww)


class _FgOffset(torch.autograd.Function):
    """First half of the FG-MSA offset head (stj_fg_offset_fwd/_bwd with fh = NULL): off = tanh(o . W1) * scale, o read in the
    offset conv's own [B,H,W,G*gc] layout, off [B,G,HW,2]."""
    @staticmethod
    def forward(ctx, o, t1, p1, scale, G):
        _req_cuda(o)
        o = o.contiguous()
        B, Hh, Ww, C = o.shape
        HW, gc = Hh * Ww, C // G
        off = torch.empty((B, G, HW, 2), dtype=o.dtype, device=o.device)
        call('stj_fg_offset_fwd', _p(o), _p(p1.c), None, None, None, _p(off), None, B, HW, G, gc, 8, float(scale), 0, _dt(o), _st())
        ctx.p1, ctx.geo = p1, (B, HW, G, gc, float(scale))
        ctx.save_for_backward(o, off)
        return off

    @staticmethod
    def backward(ctx, doff):
        o, off = ctx.saved_tensors
        B, HW, G, gc, scale = ctx.geo
        dO = torch.empty_like(o)
        call('stj_fg_offset_bwd', _p(o), _p(off), _p(ctx.p1.c), None, _p(doff.contiguous()), None, _p(dO), None, None, _p(ctx.p1.grad),
             None, None, B, HW, G, gc, 8, scale, 0, _dt(o), _st())
        return dO, None, None, None, None


def fg_offset(o, p1, scale, G):
    return _FgOffset.apply(o, p1.master, p1, scale, G)


def fgoff_ok(dtype, Hh, Ww, C, G):
    from ._lib import lib
    return bool(lib().stj_fgoff_supported(int(Hh), int(Ww), int(C), int(G), DTYPE_CODE[dtype]))


def fgoff_pack(pw, dtype, out=None):
    """MFMA-fragment-ordered copies (both directions) of FG-MSA's offset conv kernel in `dtype` (stj_fgoff_pack).  Once per step."""
    from ._lib import lib
    if out is None:
        out = torch.empty(int(lib().stj_fgoff_pack_workspace_bytes(DTYPE_CODE[dtype])), dtype=torch.uint8, device=pw.master.device)
    call('stj_fgoff_pack', _p(pw.master), _p(out), DTYPE_CODE[dtype], _st())
    return out


class _FgOffsetChain(torch.autograd.Function):
    """off [B,G,HW,2] = tanh(conv_offset(q)) * scale with conv_offset = grouped 3x3 conv -> LayerNorm(eps) -> gelu -> per-group 1x1 conv 48 -> 2
    (FG_MSA.py:84-92,109-123) as one launch per direction (csrc/fgoff_fused.hip) in place of grouped_conv3 + layernorm + gelu + fg_offset.
    ps: Params (conv kernel, conv bias, gamma, beta, proj kernel); pack: fgoff_pack()."""
    @staticmethod
    def forward(ctx, q, trig, ps, pack, scale, eps):
        from ._lib import FgOffArgs
        _req_cuda(q)
        q = q.contiguous()
        B, Hh, Ww, C = q.shape
        G, M = 8, B * Hh * Ww
        pw, pb, pg, pbe, p1 = ps
        dev, dt = q.device, q.dtype
        off = torch.empty((B, G, Hh * Ww, 2), dtype=dt, device=dev)
        train = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        sv = {}
        if train:
            sv = dict(cols=torch.empty((M, G, 9 * (C // G)), dtype=dt, device=dev), c=torch.empty((M, C), dtype=dt, device=dev),
                      mean=torch.empty((M,), dtype=torch.float32, device=dev), rstd=torch.empty((M,), dtype=torch.float32, device=dev))
        a = FgOffArgs(B=B, H=Hh, W=Ww, dtype=DTYPE_CODE[dt], scale=float(scale), eps=float(eps), q=_ip(q), pack=_ip(pack), bias=_ip(pb.master),
                      gamma=_ip(pg.master), beta=_ip(pbe.master), w1=_ip(p1.c), off=_ip(off), **{k: _ip(v) for k, v in sv.items()})
        call('stj_fgoff_fwd', ctypes.byref(a), _st())
        if train:
            ctx.ps, ctx.pack, ctx.geo = ps, pack, (B, Hh, Ww, C, G, float(scale), float(eps))
            ctx.save_for_backward(off, sv['cols'], sv['c'], sv['mean'], sv['rstd'])
        return off

    @staticmethod
    def backward(ctx, doff):
        from ._lib import FgOffArgs
        off, cols, c, mean, rstd = ctx.saved_tensors
        B, Hh, Ww, C, G, scale, eps = ctx.geo
        pw, pb, pg, pbe, p1 = ctx.ps
        M, Cg = B * Hh * Ww, C // G
        K = 9 * Cg
        doff = doff.contiguous()
        dc = torch.empty_like(c)
        dq = torch.empty((B, Hh, Ww, C), dtype=c.dtype, device=c.device)
        a = FgOffArgs(B=B, H=Hh, W=Ww, dtype=DTYPE_CODE[c.dtype], scale=scale, eps=eps, pack=_ip(ctx.pack), gamma=_ip(pg.master), beta=_ip(pbe.master),
                      w1=_ip(p1.c), off=_ip(off), c=_ip(c), mean=_ip(mean), rstd=_ip(rstd), doff=_ip(doff), dc=_ip(dc), dq=_ip(dq),
                      d_w1=_ip(p1.grad), d_gamma=_ip(pg.grad), d_beta=_ip(pbe.grad), d_bias=_ip(pb.grad))
        call('stj_fgoff_bwd', ctypes.byref(a), _st())
        # dW[k, g*cog+n] += sum_m cols[m,g,k] dc[m, g*cog+n]   (as _GroupedConv3.backward; queued on the grouped weight-gradient launch)
        gemm(cols, dc, pw.grad, K, Cg, M, (0, K, 1, G * K), (0, Cg, C, 1), (0, Cg, C), _dt(cols), nb=(1, G), c_f32=1, accumulate=1, splitk=0)
        return dq, None, None, None, None, None


def fgoff_chain(q, pw, pb, pg, pbe, p1, pack, scale, eps):
    return _FgOffsetChain.apply(q, pw.master, (pw, pb, pg, pbe, p1), pack, scale, eps)


class _FgQuery(torch.autograd.Function):
    """Second half: fh = off . W2 + b2 (1x1 conv 2 -> C2), [B,G,HW,C2]; with qres [B,HW,C2] the output is the group-major decoder
    query [G,B,HW,C2] = qres (broadcast over the groups) + fh  (modules.py:827-831), written once."""
    @staticmethod
    def forward(ctx, off, qres, t2, t3, p2, pb2):
        _req_cuda(off)
        off = off.contiguous()
        B, G, HW, _ = off.shape
        C2 = p2.c.shape[-1]
        zmajor = 1 if qres is not None else 0
        if qres is not None:
            qres = qres.contiguous()
        fh = torch.empty((G, B, HW, C2) if zmajor else (B, G, HW, C2), dtype=off.dtype, device=off.device)
        call('stj_fg_offset_fwd', None, None, _p(p2.c), _p(pb2.master), _p(qres), _p(off), _p(fh), B, HW, G, 1, C2, 1.0, zmajor,
             _dt(off), _st())
        ctx.ps, ctx.geo = (p2, pb2), (B, HW, G, C2, zmajor)
        ctx.save_for_backward(off)
        return fh

    @staticmethod
    def backward(ctx, dfh):
        (off,) = ctx.saved_tensors
        p2, pb2 = ctx.ps
        B, HW, G, C2, zmajor = ctx.geo
        doff = torch.empty_like(off)
        dq = torch.empty((B, HW, C2), dtype=off.dtype, device=off.device) if (zmajor and ctx.needs_input_grad[1]) else None
        call('stj_fg_offset_bwd', None, _p(off), None, _p(p2.c), None, _p(dfh.contiguous()), None, _p(dq), _p(doff), None, _p(p2.grad),
             _p(pb2.grad), B, HW, G, 1, C2, 1.0, zmajor, _dt(off), _st())
        return doff, dq, None, None, None, None


def fg_query(off, p2, pb2, qres=None):
    return _FgQuery.apply(off, qres, p2.master, pb2.master, p2, pb2)


# ----------------------------------------------------------------------------------------------------
# trajNet input plumbing and branch sums (one launch each instead of 3-8 torch dispatches on a chain of 5 us kernels)
# ----------------------------------------------------------------------------------------------------
def agent_prep(obs, occ, dtype):
    """obs [B,n_obs,T,8], occ [B,n_occ,T,8] -> x5 [B*A*T,5], v3 [B*A,3], vt [B*A,T] int32, cmi [B,A] int32, cmf [B,A] (no gradients)."""
    _req_cuda(obs, occ)
    obs, occ = obs.float().contiguous(), occ.float().contiguous()
    B, n_obs, Tn, _ = obs.shape
    n_occ = occ.shape[1]
    A = n_obs + n_occ
    dev = obs.device
    x5 = torch.empty((B * A * Tn, 5), dtype=dtype, device=dev)
    v3 = torch.empty((B * A, 3), dtype=dtype, device=dev)
    vt = torch.empty((B * A, Tn), dtype=torch.int32, device=dev)
    cmi = torch.empty((B, A), dtype=torch.int32, device=dev)
    cmf = torch.empty((B, A), dtype=dtype, device=dev)
    call('stj_agent_prep', _p(obs), _p(occ), n_obs, n_occ, B, Tn, _p(x5), _p(v3), _p(vt), _p(cmi), _p(cmf), DTYPE_CODE[dtype], _st())
    return x5, v3, vt, cmi, cmf


class _AgentMix(torch.autograd.Function):
    """concat = enc * cm ; qin = concat + embed  (trajNet.py:166-170)."""
    @staticmethod
    def forward(ctx, enc, embed, cm):
        _req_cuda(enc)
        enc, embed = enc.contiguous(), embed.contiguous()
        B, A, C = enc.shape
        concat, qin = torch.empty_like(enc), torch.empty_like(enc)
        call('stj_agent_mix_fwd', _p(enc), _p(embed), _p(cm), _p(concat), _p(qin), B, A, C, _dt(enc), _st())
        ctx.geo = (B, A, C, embed.shape)
        ctx.save_for_backward(cm)
        return concat, qin

    @staticmethod
    def backward(ctx, dconcat, dqin):
        (cm,) = ctx.saved_tensors
        B, A, C, eshape = ctx.geo
        dconcat = dconcat.contiguous() if dconcat is not None else None
        dqin = dqin.contiguous() if dqin is not None else None
        denc = torch.empty((B, A, C), dtype=cm.dtype, device=cm.device)
        dembed = torch.empty(eshape, dtype=cm.dtype, device=cm.device) if ctx.needs_input_grad[1] else None
        call('stj_agent_mix_bwd', _p(dconcat), _p(dqin), _p(cm), _p(denc), _p(dembed), B, A, C, _dt(cm), _st())
        return denc, dembed, None


def agent_mix(enc, embed, cm):
    return _AgentMix.apply(enc, embed, cm)


class _AgentSum(torch.autograd.Function):
    """out = enc + value + embed (trajNet.py:171)."""
    @staticmethod
    def forward(ctx, enc, value, embed):
        _req_cuda(enc)
        enc, value, embed = enc.contiguous(), value.contiguous(), embed.contiguous()
        B, A, C = enc.shape
        out = torch.empty_like(enc)
        call('stj_agent_sum_fwd', _p(enc), _p(value), _p(embed), _p(out), B, A, C, _dt(enc), _st())
        ctx.geo = (B, A, C, embed.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, A, C, eshape = ctx.geo
        dout = dout.contiguous()
        dembed = None
        if ctx.needs_input_grad[2]:
            dembed = torch.empty(eshape, dtype=dout.dtype, device=dout.device)
            call('stj_agent_sum_bwd', _p(dout), _p(dembed), B, A, C, _dt(dout), _st())
        return dout, dout, dembed


def agent_sum(enc, value, embed):
    return _AgentSum.apply(enc, value, embed)


class _AgentOut(torch.autograd.Function):
    """LayerNorm_{obs | occ}(enc + value + embed): the tail of TrajNet.call (trajNet.py:171-187) as one launch per direction -- the first n_obs
    agents of a scene go through obs_norm, the others through occ_norm."""
    @staticmethod
    def forward(ctx, enc, value, embed, g0m, b0m, g1m, b1m, pg0, pb0, pg1, pb1, n_obs, eps):
        _req_cuda(enc)
        enc, value, embed = enc.contiguous(), value.contiguous(), embed.contiguous()
        B, A, C = enc.shape
        out, y = torch.empty_like(enc), torch.empty_like(enc)
        mean = torch.empty(B * A, dtype=torch.float32, device=enc.device)
        rstd = torch.empty_like(mean)
        call('stj_agent_out_fwd', _p(enc), _p(value), _p(embed), _p(pg0.master), _p(pb0.master), _p(pg1.master), _p(pb1.master), _p(out), _p(y),
             _p(mean), _p(rstd), B, A, n_obs, C, float(eps), _dt(enc), _st())
        ctx.ps, ctx.geo = (pg0, pb0, pg1, pb1), (B, A, C, n_obs, embed.shape)
        ctx.save_for_backward(out, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        out, mean, rstd = ctx.saved_tensors
        pg0, pb0, pg1, pb1 = ctx.ps
        B, A, C, n_obs, eshape = ctx.geo
        dy = dy.contiguous()
        dout = torch.empty_like(out)
        dembed = torch.empty(eshape, dtype=out.dtype, device=out.device)
        call('stj_agent_out_bwd', _p(dy), _p(out), _p(mean), _p(rstd), _p(pg0.master), _p(pg1.master), _p(dout), _p(dembed), _p(pg0.grad),
             _p(pb0.grad), _p(pg1.grad), _p(pb1.grad), B, A, n_obs, C, _dt(out), _st())
        return (dout, dout, dembed) + (None,) * 10


def agent_out(enc, value, embed, pg0, pb0, pg1, pb1, n_obs, eps):
    return _AgentOut.apply(enc, value, embed, pg0.master, pb0.master, pg1.master, pb1.master, pg0, pb0, pg1, pb1, n_obs, eps)


# ----------------------------------------------------------------------------------------------------
# fused agent branch (csrc/agent_fused.hip): TrajEncoder for all agents / the 64-agent interaction block, one launch per direction
# ----------------------------------------------------------------------------------------------------
AGENT_PACK_KEYS = ('e_wq', 'e_wk', 'e_wv', 'e_wo', 'e_ws', 'i_wq', 'i_wk', 'i_wv', 'i_wo', 'i_w1', 'i_w2')


def _ip(t):
    """integer device address for a ctypes struct field (None -> NULL)"""
    return None if t is None else t.data_ptr()


def agent_pack(ws, dtype, out=None):
    """Transposed copies of the branch's eleven kernels in `dtype` (stj_agent_pack): what the fused forward kernels stream.  ws: dict
    AGENT_PACK_KEYS -> Param.  Once per step (the weights change)."""
    from ._lib import lib, AgentWeights
    if out is None:
        out = torch.empty(int(lib().stj_agent_pack_workspace_bytes(DTYPE_CODE[dtype])), dtype=torch.uint8, device=ws['e_wq'].master.device)
    aw = AgentWeights(*[ws[k].master.data_ptr() for k in AGENT_PACK_KEYS])
    call('stj_agent_pack', ctypes.byref(aw), _p(out), DTYPE_CODE[dtype], _st())
    return out


def agent_enc_ok(n_obs, n_occ, Tn, dtype):
    from ._lib import lib
    return bool(lib().stj_agent_enc_supported(int(n_obs), int(n_occ), int(Tn), DTYPE_CODE[dtype]))


class _AgentEnc(torch.autograd.Function):
    """enc [B,A,384], cmi [B,A] = TrajEncoder of every agent (trajNet.py:38-48,127-138) from the raw tracks obs [B,n_obs,11,8], occ
    [B,n_occ,11,8].  ws: Params {wn, bn, wv3, e_wq, e_wk, e_wv, e_wo, e_bo, e_ws, e_bs}; pack: agent_pack(); drop: (p, state, site) or None."""
    @staticmethod
    def forward(ctx, obs, occ, trig, ws, pack, dtype, drop):
        from ._lib import AgentEncArgs
        _req_cuda(obs, occ)
        obs, occ = obs.float().contiguous(), occ.float().contiguous()
        B, n_obs, Tn, _ = obs.shape
        n_occ = occ.shape[1]
        A = n_obs + n_occ
        dev = obs.device
        enc = torch.empty((B, A, 384), dtype=dtype, device=dev)
        cmi = torch.empty((B, A), dtype=torch.int32, device=dev)
        train = bool(ctx.needs_input_grad[2])
        sv = {}
        if train:
            rows = B * A * Tn
            sv = dict(s_nodes=torch.empty((rows, 64), dtype=dtype, device=dev), s_qkv=torch.empty((rows, 768), dtype=dtype, device=dev),
                      s_att=torch.empty((rows, 256), dtype=dtype, device=dev), s_pmask=torch.empty((B * A, 320), dtype=torch.int16, device=dev),
                      s_cat=torch.empty((B * A, 384), dtype=dtype, device=dev))
        p_drop, state, site = drop if drop is not None else (0.0, None, 0)
        a = AgentEncArgs(obs=_ip(obs), occ=_ip(occ), n_obs=n_obs, n_occ=n_occ, B=B, dtype=DTYPE_CODE[dtype], pack=_ip(pack),
                         wn=_ip(ws['wn'].master), bn=_ip(ws['bn'].master), wv3=_ip(ws['wv3'].master), bo=_ip(ws['e_bo'].master),
                         bs=_ip(ws['e_bs'].master), enc=_ip(enc), cmi=_ip(cmi), rng_state=_ip(state), site=site, p_drop=float(p_drop),
                         **{k: _ip(v) for k, v in sv.items()})
        call('stj_agent_enc_fwd', ctypes.byref(a), _st())
        ctx.ws, ctx.drop, ctx.geo, ctx.tdtype = ws, drop, (B, n_obs, n_occ, Tn), dtype
        if train:
            ctx.save_for_backward(obs, occ, enc, cmi, *[sv[k] for k in ('s_nodes', 's_qkv', 's_att', 's_pmask', 's_cat')])
        ctx.mark_non_differentiable(cmi)
        return enc, cmi

    @staticmethod
    def backward(ctx, denc, _dcmi):
        from ._lib import AgentEncArgs
        obs, occ, enc, cmi, s_nodes, s_qkv, s_att, s_pmask, s_cat = ctx.saved_tensors
        ws, dtype = ctx.ws, ctx.tdtype
        B, n_obs, n_occ, Tn = ctx.geo
        A = n_obs + n_occ
        rows = B * A * Tn
        dev = enc.device
        dt = DTYPE_CODE[dtype]
        denc = denc.contiguous()
        dpre_s = torch.empty((B * A, 384), dtype=dtype, device=dev)
        dout = torch.empty((rows, 320), dtype=dtype, device=dev)
        dqkv = torch.empty((rows, 768), dtype=dtype, device=dev)
        p_drop, state, site = ctx.drop if ctx.drop is not None else (0.0, None, 0)
        a = AgentEncArgs(obs=_ip(obs), occ=_ip(occ), n_obs=n_obs, n_occ=n_occ, B=B, dtype=dt, enc=_ip(enc), cmi=_ip(cmi),
                         s_nodes=_ip(s_nodes), s_qkv=_ip(s_qkv), s_att=_ip(s_att), s_pmask=_ip(s_pmask), s_cat=_ip(s_cat),
                         rng_state=_ip(state), site=site, p_drop=float(p_drop), d_enc=_ip(denc),
                         wq=_ip(ws['e_wq'].c), wk=_ip(ws['e_wk'].c), wv=_ip(ws['e_wv'].c), wo=_ip(ws['e_wo'].c), ws=_ip(ws['e_ws'].c),
                         dpre_s=_ip(dpre_s), dout=_ip(dout), dqkv=_ip(dqkv), dwn=_ip(ws['wn'].grad), dbn=_ip(ws['bn'].grad), dwv3=_ip(ws['wv3'].grad))
        call('stj_agent_enc_bwd', ctypes.byref(a), _st())
        # the three weight gradients on what the kernel wrote (queued on the grouped stream-K launch inside a model's backward pass)
        zq = ws['e_wk'].grad.data_ptr() - ws['e_wq'].grad.data_ptr()
        assert ws['e_wv'].grad.data_ptr() - ws['e_wk'].grad.data_ptr() == zq and zq % 4 == 0
        with gemm_group(not (_WG_MODE & 1)), wgrad_stream(1, s_cat, dpre_s, s_att, dout, s_nodes, dqkv):
            gemm(s_cat, dpre_s, ws['e_ws'].grad, 384, 384, B * A, (0, 0, 1, 384), (0, 0, 384, 1), (0, 0, 384), dt, c_f32=1, accumulate=1,
                 splitk=0, colsum=ws['e_bs'].grad)
            gemm(s_att, dout, ws['e_wo'].grad.view(256, 320), 256, 320, rows, (0, 0, 1, 256), (0, 0, 320, 1), (0, 0, 320), dt, c_f32=1,
                 accumulate=1, splitk=0, colsum=ws['e_bo'].grad)
            gemm(s_nodes, dqkv, ws['e_wq'].grad, 64, 64, rows, (0, 0, 1, 64), (256, 64, 768, 1), (zq // 4, 64 * 64, 64), dt, nb=(3, 4),
                 c_f32=1, accumulate=1, splitk=0)               # dW_m[h] += nodes^T dqkv[:, m, h]  (the three tfa kernels lie zq apart)
        return (None,) * 7


def agent_enc(obs, occ, ws, pack, dtype, drop=None):
    return _AgentEnc.apply(obs, occ, _trig(ws['e_ws'].master), ws, pack, dtype, drop)


def agent_int_ok(n_obs, n_occ, dtype):
    from ._lib import lib
    return bool(lib().stj_agent_int_supported(int(n_obs), int(n_occ), DTYPE_CODE[dtype]))


_ENC_SAVES = ('s_nodes', 's_qkv', 's_att', 's_pmask', 's_cat')
_INT_SAVES = ('s_concat', 's_qin', 's_q', 's_k', 's_v', 's_att', 's_v1', 's_n1', 's_h', 's_u2', 's_out')


class _AgentBranch(torch.autograd.Function):
    """key [B,64,384], cmi [B,64] = TrajNet.call (trajNet.py:125-187) from the raw tracks: the fused TrajEncoder kernel followed by the three
    interaction-block kernels (16-bit storage types); backward = three interaction kernels + the encoder kernel + nine queued weight-gradient
    products.  ws: the Params of STrajNet._agent_ws(); drop: (p, state, (site of the encoder attention, site_a, site_1, site_2)) or None."""
    @staticmethod
    def forward(ctx, obs, occ, trig, ws, pack, dtype, drop):
        from ._lib import AgentEncArgs, AgentIntArgs
        _req_cuda(obs, occ)
        obs, occ = obs.float().contiguous(), occ.float().contiguous()
        B, n_obs, Tn, _ = obs.shape
        n_occ = occ.shape[1]
        A, C = n_obs + n_occ, 384
        dev, dt = obs.device, DTYPE_CODE[dtype]
        enc = torch.empty((B, A, C), dtype=dtype, device=dev)
        key = torch.empty_like(enc)
        cmi = torch.empty((B, A), dtype=torch.int32, device=dev)
        train = bool(ctx.needs_input_grad[2])
        se, si = {}, {}
        if train:
            rows = B * A * Tn
            se = dict(s_nodes=torch.empty((rows, 64), dtype=dtype, device=dev), s_qkv=torch.empty((rows, 768), dtype=dtype, device=dev),
                      s_att=torch.empty((rows, 256), dtype=dtype, device=dev), s_pmask=torch.empty((B * A, 320), dtype=torch.int16, device=dev),
                      s_cat=torch.empty((B * A, C), dtype=dtype, device=dev))
            si = {k: torch.empty((B * A, 1536 if k == 's_h' else C), dtype=dtype, device=dev) for k in _INT_SAVES}
        p_drop, state, sites = drop if drop is not None else (0.0, None, (0, 0, 0, 0))
        m = lambda k: _ip(ws[k].master)
        a = AgentEncArgs(obs=_ip(obs), occ=_ip(occ), n_obs=n_obs, n_occ=n_occ, B=B, dtype=dt, pack=_ip(pack), wn=m('wn'), bn=m('bn'), wv3=m('wv3'),
                         bo=m('e_bo'), bs=m('e_bs'), enc=_ip(enc), cmi=_ip(cmi), rng_state=_ip(state), site=sites[0], p_drop=float(p_drop),
                         **{k: _ip(v) for k, v in se.items()})
        call('stj_agent_enc_fwd', ctypes.byref(a), _st())
        acc = torch.empty((10, B * A, C), dtype=torch.float32, device=dev)       # slabs of the six heads' / four hidden chunks' partial sums
        ai = AgentIntArgs(enc=_ip(enc), cmi=_ip(cmi), n_obs=n_obs, n_occ=n_occ, B=B, dtype=dt, pack=_ip(pack), seg=_ip(ws['seg'].c),
                          bo=m('i_bo'), g1=m('g1'), be1=m('be1'), b1=m('b1'), b2=m('b2'), g2=m('g2'), be2=m('be2'), g_obs=m('g_obs'), b_obs=m('b_obs'),
                          g_occ=m('g_occ'), b_occ=m('b_occ'), key=_ip(key), ws_v1=_ip(acc[0]), ws_u2=_ip(acc[6]), rng_state=_ip(state),
                          site_a=sites[1], site_1=sites[2], site_2=sites[3], p_drop=float(p_drop), **{k: _ip(v) for k, v in si.items()})
        call('stj_agent_int_fwd', ctypes.byref(ai), _st())
        ctx.ws, ctx.drop, ctx.geo, ctx.tdtype = ws, drop, (B, n_obs, n_occ, Tn), dtype
        if train:
            ctx.save_for_backward(obs, occ, enc, cmi, *[se[k] for k in _ENC_SAVES], *[si[k] for k in _INT_SAVES])
        ctx.mark_non_differentiable(cmi)
        return key, cmi

    @staticmethod
    def backward(ctx, dkey, _dcmi):
        from ._lib import AgentEncArgs, AgentIntArgs
        t = ctx.saved_tensors
        obs, occ, enc, cmi = t[:4]
        se = dict(zip(_ENC_SAVES, t[4:9]))
        si = dict(zip(_INT_SAVES, t[9:]))
        ws, dtype = ctx.ws, ctx.tdtype
        B, n_obs, n_occ, Tn = ctx.geo
        A, C = n_obs + n_occ, 384
        R, rows = B * A, B * A * Tn
        dev, dt = enc.device, DTYPE_CODE[dtype]
        dkey = dkey.contiguous()
        d_enc = torch.empty((7, R, C), dtype=torch.float32, device=dev)          # slabs: the residual's share + one per head
        dn1acc = torch.empty((4, R, C), dtype=torch.float32, device=dev)         # one per hidden chunk
        dY = {k: torch.empty((R, 1536 if k == 'dpre1' else C), dtype=dtype, device=dev) for k in ('dq', 'dk', 'dv', 'dv1', 'dpre1', 'dz2')}
        p_drop, state, sites = ctx.drop if ctx.drop is not None else (0.0, None, (0, 0, 0, 0))
        m, c, gr = (lambda k: _ip(ws[k].master)), (lambda k: _ip(ws[k].c)), (lambda k: _ip(ws[k].grad))
        ai = AgentIntArgs(enc=_ip(enc), cmi=_ip(cmi), n_obs=n_obs, n_occ=n_occ, B=B, dtype=dt, seg=c('seg'), g1=m('g1'), g2=m('g2'), g_obs=m('g_obs'),
                          g_occ=m('g_occ'), rng_state=_ip(state), site_a=sites[1], site_1=sites[2], site_2=sites[3], p_drop=float(p_drop),
                          dkey=_ip(dkey), wq=c('i_wq'), wk=c('i_wk'), wv=c('i_wv'), wo=c('i_wo'), w1=c('i_w1'), w2=c('i_w2'), d_enc=_ip(d_enc),
                          ws_dn1=_ip(dn1acc), dseg=gr('seg'), dg1=gr('g1'), dbe1=gr('be1'), dg2=gr('g2'), dbe2=gr('be2'), dg_obs=gr('g_obs'),
                          db_obs=gr('b_obs'), dg_occ=gr('g_occ'), db_occ=gr('b_occ'), **{k: _ip(v) for k, v in si.items()},
                          **{k: _ip(v) for k, v in dY.items()})
        call('stj_agent_int_bwd', ctypes.byref(ai), _st())
        dpre_s = torch.empty((R, C), dtype=dtype, device=dev)
        dout = torch.empty((rows, 320), dtype=dtype, device=dev)
        dqkv = torch.empty((rows, 768), dtype=dtype, device=dev)
        a = AgentEncArgs(obs=_ip(obs), occ=_ip(occ), n_obs=n_obs, n_occ=n_occ, B=B, dtype=dt, enc=_ip(enc), cmi=_ip(cmi),
                         rng_state=_ip(state), site=sites[0], p_drop=float(p_drop), d_enc=_ip(d_enc), d_enc_f32=7,
                         wq=c('e_wq'), wk=c('e_wk'), wv=c('e_wv'), wo=c('e_wo'), ws=c('e_ws'),
                         dpre_s=_ip(dpre_s), dout=_ip(dout), dqkv=_ip(dqkv), dwn=gr('wn'), dbn=gr('bn'), dwv3=gr('wv3'),
                         **{k: _ip(v) for k, v in se.items()})
        call('stj_agent_enc_bwd', ctypes.byref(a), _st())
        # the nine weight gradients on what the kernels wrote (queued on the grouped stream-K launch inside a model's backward pass)
        H, hs = 6, C // 6
        zq = ws['e_wk'].grad.data_ptr() - ws['e_wq'].grad.data_ptr()
        assert ws['e_wv'].grad.data_ptr() - ws['e_wk'].grad.data_ptr() == zq and zq % 4 == 0
        keep = list(si.values()) + list(dY.values()) + list(se.values()) + [dpre_s, dout, dqkv]
        with gemm_group(not (_WG_MODE & 1)), wgrad_stream(1, *keep):
            for x, dy, w in ((si['s_qin'], dY['dq'], 'i_wq'), (si['s_concat'], dY['dk'], 'i_wk'), (si['s_concat'], dY['dv'], 'i_wv')):
                gemm(x, dy, ws[w].grad, C, hs, R, (0, 0, 1, C), (0, hs, C, 1), (0, C * hs, hs), dt, nb=(1, H), c_f32=1, accumulate=1, splitk=0)   # dW[h] += x^T dy[:, h]
            gemm(si['s_att'], dY['dv1'], ws['i_wo'].grad.view(C, C), C, C, R, (0, 0, 1, C), (0, 0, C, 1), (0, 0, C), dt, c_f32=1, accumulate=1,
                 splitk=0, colsum=ws['i_bo'].grad)
            gemm(si['s_n1'], dY['dpre1'], ws['i_w1'].grad, C, 1536, R, (0, 0, 1, C), (0, 0, 1536, 1), (0, 0, 1536), dt, c_f32=1, accumulate=1,
                 splitk=0, colsum=ws['b1'].grad)
            gemm(si['s_h'], dY['dz2'], ws['i_w2'].grad, 1536, C, R, (0, 0, 1, 1536), (0, 0, C, 1), (0, 0, C), dt, c_f32=1, accumulate=1,
                 splitk=0, colsum=ws['b2'].grad)
            gemm(se['s_cat'], dpre_s, ws['e_ws'].grad, C, C, R, (0, 0, 1, C), (0, 0, C, 1), (0, 0, C), dt, c_f32=1, accumulate=1, splitk=0,
                 colsum=ws['e_bs'].grad)
            gemm(se['s_att'], dout, ws['e_wo'].grad.view(256, 320), 256, 320, rows, (0, 0, 1, 256), (0, 0, 320, 1), (0, 0, 320), dt, c_f32=1,
                 accumulate=1, splitk=0, colsum=ws['e_bo'].grad)
            gemm(se['s_nodes'], dqkv, ws['e_wq'].grad, 64, 64, rows, (0, 0, 1, 64), (256, 64, 768, 1), (zq // 4, 64 * 64, 64), dt, nb=(3, 4),
                 c_f32=1, accumulate=1, splitk=0)
        return (None,) * 7


def agent_branch(obs, occ, ws, pack, dtype, drop=None):
    return _AgentBranch.apply(obs, occ, _trig(ws['e_ws'].master), ws, pack, dtype, drop)


# ----------------------------------------------------------------------------------------------------
# max over time (GlobalMaxPooling1D)
# ----------------------------------------------------------------------------------------------------
class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _req_cuda(x)
        x = x.contiguous()
        outer, Tn, C = x.shape[:-2].numel(), x.shape[-2], x.shape[-1]
        y = torch.empty(x.shape[:-2] + (C,), dtype=x.dtype, device=x.device)
        idx = torch.empty(outer * C, dtype=torch.int32, device=x.device)
        call('stj_maxpool_fwd', _p(x), _p(y), _p(idx), outer, Tn, C, _dt(x), _st())
        ctx.geo = (outer, Tn, C, x.shape)
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        outer, Tn, C, shape = ctx.geo
        dy = dy.contiguous()
        dx = torch.empty(shape, dtype=dy.dtype, device=dy.device)
        call('stj_maxpool_bwd', _p(dy), _p(x), _p(y), _p(dx), outer, Tn, C, _dt(dy), _st())
        return dx


def maxpool_time(x):
    return _MaxPool.apply(x)


# ----------------------------------------------------------------------------------------------------
# patch embedding conv (4x4 stride 4) = im2col + dense
# ----------------------------------------------------------------------------------------------------
def _ln_bwd_plain(dy, x, mean, rstd, pg, pb, rows, C):
    """stj_layernorm_bwd of a plain [rows, C] norm (no gather, one parameter set): -> dx; dgamma / dbeta accumulate."""
    dx = torch.empty_like(x)
    if pg.part is not None and pb.part is not None:
        dg, db, nparts, pstride = pg.part[0], pb.part[0], pg.part[1], pg.part[2]
    else:
        dg, db, nparts, pstride = pg.grad, pb.grad, 1, 0
    call('stj_layernorm_bwd', _p(dy), _p(x), _p(pg.master), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), rows, C, 0, 0, 0, 1, 0,
         _p(None), nparts, pstride, _dt(x), _st())
    return dx


LN_CHAIN = True       # the stem's two LayerNorm backward passes as one launch (stj_layernorm_bwd_chain); False = two stj_layernorm_bwd launches


def _ln_grad_targets(pg, pb):
    """(dgamma, dbeta, nparts, part_stride) of a LayerNorm's parameters: the rotating partial copies when the model set them up."""
    if pg.part is not None and pb.part is not None:
        return pg.part[0], pb.part[0], pg.part[1], pg.part[2]
    return pg.grad, pb.grad, 1, 0


def _ln_chain_ok(C, dtype):
    from ._lib import lib
    return bool(lib().stj_layernorm_bwd_chain_supported(C, DTYPE_CODE[dtype]))


class _PatchEmbed(torch.autograd.Function):
    """PatchEmbed.call (modules.py:437-446) + what the stem does with it (modules.py:572-590), ONE launch (csrc/patch_embed.hip):
    y = LN2(LN(cols(src) @ W + b) [+ add]).  Backward: the two LayerNorm backward launches on the saved pre-norm rows and the weight
    gradient dW += cols^T dpre (queued for the grouped stream-K launch like every Dense layer's); the rasters are data, no dx."""
    @staticmethod
    def forward(ctx, src, trig, wc, gw, pbias, pg, pb, add, pg2, pb2, geo, dtype):
        _req_cuda(src)
        Cin, ch_stride, pix_stride, eps, grad_on = geo
        src = src.contiguous()
        B, H, W = src.shape[0], src.shape[1], src.shape[2]
        K, N = wc.shape
        M = B * (H // 4) * (W // 4)
        # needs_input_grad stays True under torch.no_grad() (the master weights require grad); grad mode itself is off inside forward(), so
        # the wrapper reads it and hands it in: inference saves nothing
        train = grad_on and (ctx.needs_input_grad[1] or (add is not None and ctx.needs_input_grad[7]))
        dev = src.device
        y = torch.empty((M, N), dtype=dtype, device=dev)
        cols = pre = x2 = mean = rstd = mean2 = rstd2 = None
        if train:
            cols = torch.empty((M, K), dtype=dtype, device=dev)
            pre = torch.empty((M, N), dtype=dtype, device=dev)
            mean, rstd = torch.empty(M, dtype=torch.float32, device=dev), torch.empty(M, dtype=torch.float32, device=dev)
            if pg2 is not None:
                x2 = torch.empty((M, N), dtype=dtype, device=dev)
                mean2, rstd2 = torch.empty(M, dtype=torch.float32, device=dev), torch.empty(M, dtype=torch.float32, device=dev)
        a2 = add.contiguous().view(M, N) if add is not None else None
        call('stj_patch_embed_fwd', _p(src), _p(wc), _p(pbias.master), _p(pg.master), _p(pb.master), _p(a2),
             _p(pg2.master if pg2 is not None else None), _p(pb2.master if pb2 is not None else None), _p(cols), _p(pre), _p(x2), _p(y),
             _p(mean), _p(rstd), _p(mean2), _p(rstd2), B, H, W, Cin, pix_stride, ch_stride, N, float(eps), DTYPE_CODE[dtype], _st())
        ctx.p = (gw, pbias, pg, pb, pg2, pb2)
        ctx.has_add, ctx.add_shape = add is not None, (add.shape if add is not None else None)
        ctx.save_for_backward(cols, pre, x2, mean, rstd, mean2, rstd2)
        return y

    @staticmethod
    def backward(ctx, dy):
        cols, pre, x2, mean, rstd, mean2, rstd2 = ctx.saved_tensors
        gw, pbias, pg, pb, pg2, pb2 = ctx.p
        M, N = pre.shape
        K = cols.shape[1]
        dy = dy.contiguous().view(M, N)
        if pg2 is not None and LN_CHAIN and _ln_chain_ok(N, dy.dtype):
            # both LayerNorm backward passes in one launch (the gradient of `add` is the intermediate, written only when there is an `add`)
            dpre = torch.empty_like(pre)
            d2 = torch.empty_like(pre) if ctx.has_add else None
            (g2, b2, n2, s2), (g1, b1, n1, s1) = _ln_grad_targets(pg2, pb2), _ln_grad_targets(pg, pb)
            call('stj_layernorm_bwd_chain', _p(dy), _p(x2), _p(pg2.master), _p(mean2), _p(rstd2), _p(pre), _p(pg.master), _p(mean), _p(rstd),
                 _p(d2), _p(dpre), _p(g2), _p(b2), _p(g1), _p(b1), M, N, n2, s2, n1, s1, _dt(pre), _st())
            dadd = d2.view(ctx.add_shape) if ctx.has_add else None
        else:
            if pg2 is not None:
                dy = _ln_bwd_plain(dy, x2, mean2, rstd2, pg2, pb2, M, N)         # gradient of x2 = LN(pre) + add
            dadd = dy.view(ctx.add_shape) if ctx.has_add else None
            dpre = _ln_bwd_plain(dy, pre, mean, rstd, pg, pb, M, N)
        with wgrad_stream(1, cols, dpre):
            gemm(cols, dpre, gw, K, N, M, (0, 0, 1, K), (0, 0, N, 1), (0, 0, N), _dt(cols), c_f32=1, accumulate=1, splitk=0,
                 colsum=pbias.grad)                                                # dW += cols^T dpre ; db += 1^T dpre
        return (None,) * 7 + (dadd,) + (None,) * 4


def patch_embed_ok(Cin, Cout, dtype):
    from ._lib import lib
    return bool(lib().stj_patch_embed_supported(Cin, Cout, DTYPE_CODE[dtype]))


def patch_embed(src, pw, pbias, pg, pb, Cin, ch_stride, pix_stride, dtype, eps=1e-5, add=None, pg2=None, pb2=None):
    """src: f32 raster [B,H,W,*] -> [B*(H/4)*(W/4), Cout] tokens: LN2(LN(conv4x4s4(src)) [+ add]) (LN2 only with pg2 / pb2)."""
    N = pw.c.shape[-1]
    return _PatchEmbed.apply(src, pw.master, pw.c.view(-1, N), pw.grad.view(-1, N), pbias, pg, pb, add, pg2, pb2,
                             (Cin, ch_stride, pix_stride, eps, torch.is_grad_enabled()), dtype)


def patch_im2col(src, Cin, ch_stride, pix_stride, dtype):
    """src: f32 tensor viewed as [B,H,W,*]; returns [B*(H/4)*(W/4), 16*Cin] in `dtype` (no grad: inputs are data)."""
    _req_cuda(src)
    src = src.contiguous()
    B, H, W = src.shape[0], src.shape[1], src.shape[2]
    out = torch.empty((B * (H // 4) * (W // 4), 16 * Cin), dtype=dtype, device=src.device)
    call('stj_im2col_patch', _p(src), _p(out), B, H, W, Cin, pix_stride, ch_stride, DTYPE_CODE[dtype], _st())
    return out


# ----------------------------------------------------------------------------------------------------
# grouped 3x3 SAME conv (FG-MSA offsets): im2col -> per-group GEMM
# ----------------------------------------------------------------------------------------------------
class _GroupedConv3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_master, b_master, pw, pb, G):
        _req_cuda(x)
        x = x.contiguous()
        N, H, W, C = x.shape
        Cg = C // G
        K = 9 * Cg
        Co = pw.c.shape[-1]
        cog = Co // G
        dt = _dt(x)
        M = N * H * W
        cols = torch.empty((M, G, K), dtype=x.dtype, device=x.device)
        call('stj_im2col3', _p(x), _p(cols), N, H, W, G, Cg, dt, _st())
        y = torch.empty((N, H, W, Co), dtype=x.dtype, device=x.device)
        # group g: A = cols[:, g, :] ; B[k, n] = W_flat[k, g*cog + n] (row stride Co)
        gemm(cols, pw.c, y, M, cog, K, (0, K, G * K, 1), (0, cog, Co, 1), (0, cog, Co), dt, bias=pb.master, sBias=(0, cog), nb=(1, G))
        ctx.pw, ctx.pb, ctx.geo = pw, pb, (N, H, W, G, Cg, K, Co, cog, M)
        ctx.save_for_backward(cols)
        return y

    @staticmethod
    def backward(ctx, dy):
        (cols,) = ctx.saved_tensors
        N, H, W, G, Cg, K, Co, cog, M = ctx.geo
        pw, pb = ctx.pw, ctx.pb
        dt = _dt(cols)
        dy = dy.contiguous()
        dcols = torch.empty_like(cols)
        # dcols[:, g, k] = sum_n dy[:, g*cog+n] W[k, g*cog+n]
        gemm(dy, pw.c, dcols, M, K, cog, (0, cog, Co, 1), (0, cog, 1, Co), (0, K, G * K), dt, nb=(1, G))
        dx = torch.empty((N, H, W, G * Cg), dtype=dy.dtype, device=dy.device)
        call('stj_col2im3', _p(dcols), _p(dx), N, H, W, G, Cg, dt, _st())
        # dW[k, g*cog+n] += sum_m cols[m,g,k] dy[m, g*cog+n]
        gemm(cols, dy, pw.grad, K, cog, M, (0, K, 1, G * K), (0, cog, Co, 1), (0, cog, Co), dt, nb=(1, G), c_f32=1,
             accumulate=1, splitk=0)
        call('stj_colsum', _p(dy), _p(pb.grad), M, Co, Co, dt, _st())
        return dx, None, None, None, None, None


def grouped_conv3(x, pw, pb, G):
    return _GroupedConv3.apply(x, pw.master, pb.master, pw, pb, G)


# ----------------------------------------------------------------------------------------------------
# decoder: nearest-2x upsample folded 3x3 conv + bias + ELU
# ----------------------------------------------------------------------------------------------------
_DB_PARTS = 32


def _upconv_backward_tail(ctx, x, dpre, wd, need_dx):
    """Input gradient (dgrad kernel) and weight / bias gradient (side stream) of an up-conv from the pre-activation gradient."""
    F_, Hi, Wi, Cin, Cout = ctx.geo
    dt = _dt(x)
    dx = None
    if need_dx:
        dx = torch.empty_like(x)
        call('stj_upconv_dgrad', _p(dpre), _p(wd), _p(dx), _p(x) if ctx.x_is_elu_out else None, F_, Hi, Wi, Cin, Cout, dt, _st())
    pw, pb = ctx.pw, ctx.pb

    def wg():
        dweff = zeros_f32(16 * Cout * Cin, x.device)     # the 16 folded tap matrices
        if pb.part is not None:  # model-owned bias-gradient copies, folded into .grad once per step
            dbp, nparts, own = pb.part[0], pb.part[1], False
        else:                    # (~1000 workgroups would queue on Cout addresses otherwise)
            dbp, nparts, own = torch.zeros(_DB_PARTS * Cout, dtype=torch.float32, device=x.device), _DB_PARTS, True
        # alone on the GPU (serial mode) the large weight-gradient launches take all CUs; in the concurrent step half of them
        call('stj_upconv_wgrad', _p(x), _p(dpre), _p(dweff), _p(dbp), nparts, F_, Hi, Wi, Cin, Cout, 256 if _SERIAL else UPWG_BUDGET, dt, _st())
        call('stj_upconv_fold', _p(dweff), _p(pw.grad), Cin, Cout, _st())
        if own:
            pb.grad.add_(dbp.view(nparts, Cout).sum(0))
    if ctx.defer and not _SERIAL and (_WG_MODE & 2):
        _UPWG['items'].append((wg, x, dpre))             # launched by flush_upconv_wgrads() (the model's flush point)
    else:
        with wgrad_stream(2, x, dpre):
            wg()
    return dx


# Deferred up-conv weight gradients.  Launched where they are produced, the six weight-gradient kernels of the decoder (1.1 ms of work) share
# HBM with the input-gradient chain they run next to (the 96 <- 48 dgrad takes 0.48 ms in the step, 0.31 ms alone) and are finished long before
# anybody needs them.  The model puts a flush point behind the decoder (in backward order): the kernels are queued there, on the side stream,
# under the cross-attention / FG-MSA backward -- a chain of short launches that leaves most of the GPU idle.
# Whether an up-conv defers is decided when its FORWARD runs (ctx.defer) and only between wgrad_flush_point() and wgrad_defer_end():
# an up-conv applied outside that region (op-level use, another graph) launches its weight gradient at once, whatever ran before.
_UPWG = {'on': False, 'items': []}


# Issue order of the deferred launches.  'bwd' = the order backward produced them (full-resolution layers first, the two wide layers
# 192 -> 128 / 384 -> 192 last: those are 1024-workgroup non-persistent launches and land on Swin stage 2's backward); 'wide' = widest
# Cin first, so that they run beside the thin FG-MSA / agent chain and only the budgeted persistent launches reach into the encoder.
UPWG_ORDER = 'wide'     # round 6, alternating same-box runs: bwd 1350 / 1363 / 1349, wide 1372 / 1365 / 1359, rev 1340 / 1351 / 1353 scenes/s


# (Releasing the OLDEST deferred launches -- the 256 x 256 level's -- already at the two-skip level's junction, beside the junction and the two wide input-gradient
#  kernels that run alone there: 1 / 2 / 4 launches early 1352 / 1362 / 1355 scenes/s against 1380 with all of them deferred, profiles/r06_t_ab_upwg_early.txt.)
def flush_upconv_wgrads():
    items, _UPWG['items'] = _UPWG['items'], []
    if not items:
        return
    if UPWG_ORDER == 'wide':
        items = sorted(items, key=lambda it: -it[1].shape[-1])        # (stable: equal widths keep their backward order)
    elif UPWG_ORDER == 'rev':
        items = items[::-1]
    with wgrad_stream(2, *[t for it in items for t in it[1:]]):
        for wg, _, _ in items:
            wg()


class _WgradFlushPoint(torch.autograd.Function):
    """Identity.  Everything downstream of it in the forward pass has finished its backward when this node's backward runs."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        flush_upconv_wgrads()
        if WGRAD_SK_POINTS:
            wgrad_queue_flush(g.device)     # the decoder's dense weight gradients (the three time-kernel skips)
        return g


def wgrad_flush_point(x):
    """Mark x as the input of the region whose up-conv weight gradients are deferred (no-op without autograd).
    The region ends at wgrad_defer_end()."""
    _UPWG['on'] = DEFER_UPWG and x.requires_grad and torch.is_grad_enabled()
    _UPWG['items'] = []
    return _WgradFlushPoint.apply(x) if _UPWG['on'] else x


def wgrad_defer_end():
    """End of the forward region opened by wgrad_flush_point(): up-convs applied from here on do not defer."""
    _UPWG['on'] = False


DEFER_UPWG = True


class _UpConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_master, b_master, pw, pb, grad_is_pre, x_is_elu_out, prep):
        _req_cuda(x)
        x = x.contiguous()
        F_, Hi, Wi, Cin = x.shape
        Cout = pw.master.shape[-1]
        dt = _dt(x)
        if prep is not None:         # folded tap matrices made ahead of time (the model folds all decoder weights at the start of a step)
            wf, wd = prep
        else:
            wf, wd = upconv_prep(pw, x.dtype)
        y = torch.empty((F_, 2 * Hi, 2 * Wi, Cout), dtype=x.dtype, device=x.device)
        call('stj_upconv_fwd', _p(x), _p(wf), _p(pb.master), _p(y), F_, Hi, Wi, Cin, Cout, ACT_ELU, dt, _st())
        ctx.pw, ctx.pb, ctx.geo = pw, pb, (F_, Hi, Wi, Cin, Cout)
        ctx.grad_is_pre, ctx.x_is_elu_out = grad_is_pre, x_is_elu_out
        ctx.defer = _UPWG['on']
        ctx.save_for_backward(x, y, wd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, wd = ctx.saved_tensors
        F_, Hi, Wi, Cin, Cout = ctx.geo
        dt = _dt(x)
        dy = dy.contiguous()
        if ctx.grad_is_pre:          # the consumer already folded ELU'(y) into the gradient it returned
            dpre = dy
        else:
            dpre = torch.empty_like(dy)
            call('stj_unary_bwd', _p(dy), _p(y), _p(dpre), dy.numel(), U_ELU, 0.0, dt, _st())
        dx = _upconv_backward_tail(ctx, x, dpre, wd, ctx.needs_input_grad[0])
        return dx, None, None, None, None, None, None, None


class _UpConvAdd(torch.autograd.Function):
    """y = ELU(upconv(x)) + r1 [, y2 = y + r2] with the sums in the up-conv's epilogue (stj_upconv_fwd_res): the decoder skips of
    modules.py:750-765.  The ELU output itself is never stored; backward recovers ELU' from y - r1 (stj_elu_res_bwd), which also adds
    the two incoming gradients when there are two outputs."""
    @staticmethod
    def forward(ctx, x, r1, r2, w_master, b_master, pw, pb, prep):
        _req_cuda(x, r1)
        x, r1 = x.contiguous(), r1.contiguous()
        F_, Hi, Wi, Cin = x.shape
        Cout = pw.master.shape[-1]
        dt = _dt(x)
        wf, wd = prep if prep is not None else upconv_prep(pw, x.dtype)
        y = torch.empty((F_, 2 * Hi, 2 * Wi, Cout), dtype=x.dtype, device=x.device)
        y2 = None
        if r2 is not None:
            r2 = r2.contiguous()
            y2 = torch.empty_like(y)
        call('stj_upconv_fwd_res', _p(x), _p(wf), _p(pb.master), _p(y), _p(r1), _p(y2), _p(r2), F_, Hi, Wi, Cin, Cout, dt, _st())
        ctx.save_for_backward(x, y, r1, wd)
        ctx.pw, ctx.pb, ctx.geo = pw, pb, (F_, Hi, Wi, Cin, Cout)
        ctx.x_is_elu_out, ctx.two = False, r2 is not None
        ctx.defer = _UPWG['on']
        return (y, y2) if r2 is not None else y

    @staticmethod
    def backward(ctx, dy, dy2=None):
        x, y, r1, wd = ctx.saved_tensors
        dt = _dt(x)
        dy = dy.contiguous()          # (undefined output gradients arrive materialised as zeros: dy is never None)
        dpre = torch.empty_like(dy)
        gsum = None
        if dy2 is not None:
            dy2 = dy2.contiguous()
            gsum = torch.empty_like(dy)
        call('stj_elu_res_bwd', _p(dy), _p(dy2), _p(y), _p(r1), _p(dpre), _p(gsum), dy.numel(), dt, _st())
        dx = _upconv_backward_tail(ctx, x, dpre, wd, ctx.needs_input_grad[0])
        dr1 = gsum if gsum is not None else dy
        return dx, dr1, (dy2 if ctx.two else None), None, None, None, None, None


# The skip sums ride in the up-conv epilogue in INFERENCE only.  Measured at B=8 bf16: the training step is 1.2 % SLOWER with them there
# (901 vs 912 scenes/s: a workgroup owns 32 of the 128 couts, so the skip operands are read and the sums written in 64-byte pieces, 138 vs
# 92 us for the 192 -> 128 layer, while the separate adds stream whole lines at 5 TB/s; and as one separate fused pass 0.5 % slower), the
# B=32 fp16 forward 1.4 % faster.


SKIP_BWD_FUSED = True      # (False: the level's backward junction as an elementwise add + the ELU' pass; the f32 mode always)
SKIP_JUNCTION = True       # the junction also applies the SKIPS' ELU' (stj_skip_junction_bwd); False: one stj_unary_bwd per skip, as up to round 5


class _UpConvSkips(torch.autograd.Function):
    """y1 = ELU(upconv(x)) + r1 [, y2 = y1 + r2] (a decoder level with its skips, modules.py:750-765) with the forward as the launches that
    measure fastest in training (the up-conv, then the elementwise adds) and the backward's junction as ONE launch:
      skips_pre = False  the sum of the two incoming gradients and its product with ELU'(ELU output) (stj_elu_res_bwd with r = NULL) instead of an
                         elementwise add followed by the ELU' pass (two skips only);
      skips_pre = True   r1 / r2 are ELU outputs of producers called with grad_is_pre = True (linear_z): stj_skip_junction_bwd also returns THEIR
                         gradients times ELU'(r) -- the three (one skip: two) ELU' passes of the level in the one launch."""
    @staticmethod
    def forward(ctx, x, r1, r2, w_master, b_master, pw, pb, prep, skips_pre):
        _req_cuda(x, r1)
        x = x.contiguous()
        F_, Hi, Wi, Cin = x.shape
        Cout = pw.master.shape[-1]
        wf, wd = prep if prep is not None else upconv_prep(pw, x.dtype)
        y = torch.empty((F_, 2 * Hi, 2 * Wi, Cout), dtype=x.dtype, device=x.device)
        call('stj_upconv_fwd', _p(x), _p(wf), _p(pb.master), _p(y), F_, Hi, Wi, Cin, Cout, ACT_ELU, _dt(x), _st())
        y1 = y + r1
        y2 = y1 + r2 if r2 is not None else None
        ctx.pw, ctx.pb, ctx.geo = pw, pb, (F_, Hi, Wi, Cin, Cout)
        ctx.x_is_elu_out = False
        ctx.defer = _UPWG['on']
        ctx.skips_pre, ctx.two = skips_pre, r2 is not None
        if skips_pre:
            ctx.save_for_backward(x, y, wd, r1.contiguous(), r2.contiguous() if r2 is not None else None)
        else:
            ctx.save_for_backward(x, y, wd)
        return (y1, y2) if r2 is not None else y1

    @staticmethod
    def backward(ctx, dy1, dy2=None):
        x, y, wd = ctx.saved_tensors[:3]
        dy1 = dy1.contiguous()
        dy2 = dy2.contiguous() if ctx.two else None
        dpre = torch.empty_like(dy1)
        if ctx.skips_pre:
            r1, r2 = ctx.saved_tensors[3:]
            dr1 = torch.empty_like(dy1)
            dr2 = torch.empty_like(dy1) if ctx.two else None
            call('stj_skip_junction_bwd', _p(dy1), _p(dy2), _p(y), _p(r1), _p(r2), _p(dpre), _p(dr1), _p(dr2), dy1.numel(), _dt(x), _st())
        else:
            dr1, dr2 = torch.empty_like(dy1), dy2
            call('stj_elu_res_bwd', _p(dy1), _p(dy2), _p(y), _p(None), _p(dpre), _p(dr1), dy1.numel(), _dt(x), _st())
        dx = _upconv_backward_tail(ctx, x, dpre, wd, ctx.needs_input_grad[0])
        return dx, dr1, dr2, None, None, None, None, None, None


FUSED_SKIP_TRAIN = False       # (tests flip it: the fused form's backward, stj_elu_res_bwd, is the one a fine-tuning caller of the inference graph gets)


def skips_pre_ok(dtype):
    """Will upconv_add(skips_pre=True) take the gradients' ELU' products on itself (so the skips' producers are to be called with
    grad_is_pre=True)?  The training form of the 16-bit modes; decided HERE for both sides of the contract."""
    return SKIP_BWD_FUSED and SKIP_JUNCTION and not FUSED_SKIP_TRAIN and dtype != torch.float32 and torch.is_grad_enabled()


def upconv_add(x, pw, pb, r1, r2=None, prep=None, skips_pre=False):
    """ELU(upconv(x)) + r1 -> y, and y + r2 -> y2 when r2 is given (returns y or (y, y2)).  Fused into the up-conv epilogue for the
    16-bit wide layers (Cin = 192, 384); otherwise the up-conv followed by elementwise adds.
    skips_pre=True (only where skips_pre_ok()): r1 / r2 are ELU outputs whose producers were told grad_is_pre=True."""
    Cin, Cout = pw.master.shape[2], pw.master.shape[3]
    oshape = (x.shape[0], 2 * x.shape[1], 2 * x.shape[2], Cout)
    if skips_pre:
        if not skips_pre_ok(x.dtype):
            raise RuntimeError('upconv_add(skips_pre=True) outside skips_pre_ok(): the skips would lose their ELU\' factor')
        return _UpConvSkips.apply(x, r1.view(oshape), None if r2 is None else r2.view(oshape), pw.master, pb.master, pw, pb, prep, True)
    if (FUSED_SKIP_TRAIN or not torch.is_grad_enabled()) and x.dtype != torch.float32 and Cin > 128 and Cin % 32 == 0 and Cout % 32 == 0 and os.environ.get('STJ_NO_WS') != '1':
        return _UpConvAdd.apply(x, r1.view(oshape), None if r2 is None else r2.view(oshape), pw.master, pb.master, pw, pb, prep)
    if SKIP_BWD_FUSED and r2 is not None and x.dtype != torch.float32 and torch.is_grad_enabled():
        return _UpConvSkips.apply(x, r1.view(oshape), r2.view(oshape), pw.master, pb.master, pw, pb, prep, False)
    y = upconv(x, pw, pb, prep=prep)
    y = y + r1.view(y.shape)
    return y if r2 is None else (y, y + r2.view(y.shape))


def upconv_prep(pw, dtype):
    """Fold the 3x3 kernel [3,3,Cin,Cout] (f32 master) into the 16 effective 2x2-tap matrices of the 4 output phases, in the
    activation dtype: wf [16,Cout,Cin] for the forward / weight-gradient kernels, wd [16,Cin,Cout] for the input gradient."""
    Cin, Cout = pw.master.shape[2], pw.master.shape[3]
    wf = torch.empty((16, Cout, Cin), dtype=dtype, device=pw.master.device)
    wd = torch.empty((16, Cin, Cout), dtype=dtype, device=pw.master.device)
    call('stj_upconv_prep', _p(pw.master), _p(wf), _p(wd), Cin, Cout, DTYPE_CODE[dtype], _st())
    return wf, wd


def upconv(x, pw, pb, grad_is_pre=False, x_is_elu_out=False, prep=None):
    """x [F,Hi,Wi,Cin] -> ELU(conv3x3(upsample2(x)) + b) [F,2Hi,2Wi,Cout].
    grad_is_pre=True: contract with the (single) consumer of the output -- it returns the gradient already multiplied by
    ELU'(y) (outconv_pair / upconv with x_is_elu_out=True), so the separate ELU' pass over the largest tensors is skipped.
    x_is_elu_out=True: x is the ELU output of a producer called with grad_is_pre=True; dx is returned times ELU'(x)."""
    return _UpConv.apply(x, pw.master, pb.master, pw, pb, grad_is_pre, x_is_elu_out, prep)


def _outconv_workspace(device):
    """Caller-owned scratch of stj_outconv_bwd (per-block dW/db partials; reused by both heads in stream order)."""
    return _workspace(device, 'stj_outconv_bwd_workspace_bytes')


PAIR_OUTCONV = os.environ.get('STJ_NO_WS') != '1'     # (the paired kernel belongs to the MFMA / weight-stationary family)


class _OutConvPair(torch.autograd.Function):
    """Two 3x3 C->2 heads written straight into the [B,H,W,32] f32 model output (channel 4t+{0,1} and 4t+{2,3})."""
    @staticmethod
    def forward(ctx, xo, xf, w1m, b1m, w2m, b2m, p1w, p1b, p2w, p2b, B, Tn, t_major, x_is_elu_out, side=None):
        ctx.side = side
        _req_cuda(xo, xf)
        xo, xf = xo.contiguous(), xf.contiguous()
        F_, H, W, C = xo.shape
        out = torch.empty((B, H, W, 4 * Tn), dtype=torch.float32, device=xo.device)
        dt = _dt(xo)
        ybs, yts, yps = H * W * 4 * Tn, 4, 4 * Tn
        inner = Tn
        if t_major:            # frames ordered f = t*B + b: the kernel's (f / inner, f % inner) split then yields (t, b)
            ybs, yts, inner = 4, H * W * 4 * Tn, B
        if PAIR_OUTCONV and dt != 0 and C == 48 and Tn == 8 and H % 16 == 0 and W % 16 == 0:
            call('stj_outconv_pair_fwd', _p(xo), _p(xf), _p(p1w.master), _p(p2w.master), _p(p1b.master), _p(p2b.master), _p(out),
                 B, Tn, H, W, C, int(bool(t_major)), dt, _st())
        else:
            call('stj_outconv_fwd', _p(xo), _p(p1w.master), _p(p1b.master), vp(out.data_ptr()), F_, H, W, C, inner, ybs, yts, yps, dt, _st())
            call('stj_outconv_fwd', _p(xf), _p(p2w.master), _p(p2b.master), vp(out.data_ptr() + 8), F_, H, W, C, inner, ybs, yts, yps, dt, _st())
        ctx.ps = (p1w, p1b, p2w, p2b)
        ctx.geo = (F_, H, W, C, inner, ybs, yts, yps)
        ctx.elu_in = int(bool(x_is_elu_out))
        ctx.save_for_backward(xo, xf)
        return out

    @staticmethod
    def backward(ctx, dout):
        xo, xf = ctx.saved_tensors
        p1w, p1b, p2w, p2b = ctx.ps
        F_, H, W, C, Tn, ybs, yts, yps = ctx.geo
        dout = dout.contiguous().float()
        dt = _dt(xo)
        dxo, dxf = torch.empty_like(xo), torch.empty_like(xf)
        ws = _outconv_workspace(xo.device)
        # (round 5: leaving the partial sums' reduction to two launches of their own measured inside the noise on the main stream, 1288 vs
        #  1285 scenes/s, and -9 % on the weight-gradient side stream -- the fork at the head of backward reorders the graph's branches)
        call('stj_outconv_bwd', _p(xo), _p(p1w.master), vp(dout.data_ptr()), _p(dxo), _p(p1w.grad), _p(p1b.grad), F_, H, W, C, Tn,
             ybs, yts, yps, ctx.elu_in, _p(ws), ws.numel(), dt, _st())
        side = ctx.side if (OUTCONV_BWD_TWO_STREAMS and not _SERIAL) else None
        if side is not None:
            # the second head on the stream its branch's backward continues on (the model's second side stream): the first branch's input
            # gradient then starts behind ITS head instead of behind both
            main = torch.cuda.current_stream(xo.device)
            side.wait_stream(main)
            ws2 = _workspace(xo.device, 'stj_outconv_bwd_workspace_bytes', 1)
            with torch.cuda.stream(side):
                call('stj_outconv_bwd', _p(xf), _p(p2w.master), vp(dout.data_ptr() + 8), _p(dxf), _p(p2w.grad), _p(p2b.grad), F_, H, W, C, Tn,
                     ybs, yts, yps, ctx.elu_in, _p(ws2), ws2.numel(), dt, _st())
            for t in (dxf, dout, xf):
                t.record_stream(side)
        else:
            call('stj_outconv_bwd', _p(xf), _p(p2w.master), vp(dout.data_ptr() + 8), _p(dxf), _p(p2w.grad), _p(p2b.grad), F_, H, W, C, Tn,
                 ybs, yts, yps, ctx.elu_in, _p(ws), ws.numel(), dt, _st())
        return (dxo, dxf) + (None,) * 13


def upconv_head_ok(Hi, Wi, pw, dtype, Tn):
    """True when the inference form of the last level (kernel pw) applies to an input [F,Hi,Wi,96]: no autograd, 16-bit, 96 -> 48, whole
    8 x 16 tiles, 8 waypoints."""
    return (not torch.is_grad_enabled() and dtype != torch.float32 and tuple(pw.master.shape[2:]) == (96, 48) and Hi % 8 == 0 and Wi % 16 == 0
            and Tn == 8 and os.environ.get('STJ_NO_WS') != '1')


HEAD_CZ = 20       # channels of the projected tensor z: 9 taps x 2 outputs + 2 zero channels (csrc/conv_ws.hip HEAD_CZ)


def upconv_head(x, pw, pb, phead, prep=None):
    """Inference only: ELU(up-conv 96 -> 48) projected onto the 3x3 48 -> 2 head kernel `phead` inside the up-conv's epilogue ->
    z [F,2Hi,2Wi,HEAD_CZ] (stj_upconv_fwd_head); the [F,2Hi,2Wi,48] tensor is never written."""
    _req_cuda(x)
    F_, Hi, Wi, Cin = x.shape
    wf, _ = prep if prep is not None else upconv_prep(pw, x.dtype)
    z = torch.empty((F_, 2 * Hi, 2 * Wi, HEAD_CZ), dtype=x.dtype, device=x.device)
    call('stj_upconv_fwd_head', _p(x.contiguous()), _p(wf), _p(pb.master), _p(phead.master), _p(z), F_, Hi, Wi, Cin, 48, _dt(x), _st())
    return z


def heads_gather(zo, zf, p1b, p2b, B, Tn, t_major=False):
    """[B,H,W,4*Tn] f32 = bias + the 9-neighbour sums of the two projected tensors (stj_outconv_pair_gather)."""
    F_, H, W, _ = zo.shape
    out = torch.empty((B, H, W, 4 * Tn), dtype=torch.float32, device=zo.device)
    call('stj_outconv_pair_gather', _p(zo), _p(zf), _p(p1b.master), _p(p2b.master), _p(out), B, Tn, H, W, 1 if t_major else 0, _dt(zo), _st())
    return out


OUTCONV_BWD_TWO_STREAMS = True     # round 6, alternating same-box runs: 1383 / 1383 / 1382 / 1385 against 1376 / 1376 / 1379 / 1365 scenes/s on one stream


def outconv_pair(xo, xf, p1w, p1b, p2w, p2b, B, Tn, t_major=False, x_is_elu_out=False, side=None):
    """side: the stream the second branch (xf) was computed on, when the caller runs the two branches on two streams."""
    return _OutConvPair.apply(xo, xf, p1w.master, p1b.master, p2w.master, p2b.master, p1w, p1b, p2w, p2b, B, Tn, t_major, x_is_elu_out, side)


# ----------------------------------------------------------------------------------------------------
# loss
# ----------------------------------------------------------------------------------------------------
class _OgmFlowLoss(torch.autograd.Function):
    """-> (observed_xe, occluded_xe, flow, flow_warp_xe, total): five 0-dim tensors.  `total` (their sum, train.py:221) is an output
    of its own so that a step which only differentiates the sum runs no select / add / zeros glue around the two loss kernels."""
    @staticmethod
    def forward(ctx, logits, gt_obs, gt_occ, gt_flow, origin, gate, ogm_w, occ_w, fow, replica, flags, coef_pre=None, unit=None, fin_stream=None):
        _req_cuda(logits)
        logits = logits.contiguous().float()
        B, H, W, _ = logits.shape
        dev = logits.device
        loss = torch.empty(5, dtype=torch.float32, device=dev)      # the four terms + their sum
        coef = torch.empty(32, dtype=torch.float32, device=dev)
        ctx.unit_grad = None
        if coef_pre is not None and unit is not None and LOSS_FUSED_BWD:
            # the caller announced the unit gradient it will differentiate `total` with, and the backward coefficients are there already
            # (loss_coef: they depend on the ground truth alone): forward sums and d(total)/d(logits) in one pass over the logits
            sums = zeros_f32(128 * 40, dev)
            dlogits = torch.empty_like(logits)
            w = (B, H, W, float(ogm_w), float(occ_w), float(fow), float(replica), int(flags))
            ctx.fin_stream = fin_stream if (LOSS_FIN_SIDE and not _SERIAL) else None
            if ctx.fin_stream is None:
                call('stj_loss_fwd_bwd', _p(logits), _p(gt_obs), _p(gt_occ), _p(gt_flow), _p(origin), _p(gate), _p(coef_pre), _p(sums), _p(loss),
                     _p(coef), _p(dlogits), *w, _st())
            else:
                # the loss VALUES are read by nobody on the backward path: their 1-workgroup finalize launch goes behind the pass on the
                # caller's side stream (the caller joins it: GraphedTrainStep, after backward) instead of in front of the first backward kernel
                call('stj_loss_fwd_bwd', _p(logits), _p(gt_obs), _p(gt_occ), _p(gt_flow), _p(origin), _p(gate), _p(coef_pre), _p(sums), None,
                     None, _p(dlogits), *w, _st())
                ctx.fin_stream.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(ctx.fin_stream):
                    call('stj_loss_finalize', _p(sums), _p(gate), _p(loss), _p(coef), *w, _st())
                for t in (sums, gate, loss, coef):
                    t.record_stream(ctx.fin_stream)
            ctx.unit_grad, ctx.dlogits = unit, dlogits
        else:
            sums = zeros_f32(32 * 40, dev)       # 32 copies of the 40 accumulators (stj_loss_fwd)
            call('stj_loss_fwd', _p(logits), _p(gt_obs), _p(gt_occ), _p(gt_flow), _p(origin), _p(gate), _p(sums), _p(loss), _p(coef),
                 B, H, W, float(ogm_w), float(occ_w), float(fow), float(replica), int(flags), _st())
        ctx.geo = (B, H, W, int(flags))
        ctx.set_materialize_grads(False)     # a step that differentiates only `total` gets None for the four terms: one-scalar path below
        ctx.save_for_backward(logits, gt_obs, gt_occ, gt_flow, origin, coef)
        terms = loss[:4]
        ctx.mark_non_differentiable(terms)
        return tuple(terms.unbind(0)) + (loss[4], terms)

    @staticmethod
    def backward(ctx, g0, g1, g2, g3, gt, _gvec):
        logits, gt_obs, gt_occ, gt_flow, origin, coef = ctx.saved_tensors
        B, H, W, flags = ctx.geo
        parts = (g0, g1, g2, g3)
        if all(g is None for g in parts) and gt is None:
            return (None,) * 14
        if ctx.unit_grad is not None and ctx.dlogits is not None and all(g is None for g in parts) and gt.data_ptr() == ctx.unit_grad.data_ptr():
            LOSS_FUSED_STATS['hits'] += 1          # the announced unit gradient: d/dlogits was written by the forward pass
            dl, ctx.dlogits = ctx.dlogits, None
            return (dl,) + (None,) * 13
        if ctx.unit_grad is not None:              # (also a second backward through a retained graph: the stored gradient was handed over once)
            LOSS_FUSED_STATS['misses'] += 1        # some other upstream gradient: the general kernel (the forward's d/dlogits is dropped)
            ctx.dlogits = None
            if ctx.fin_stream is not None:         # (its coefficients come from the finalize launch)
                torch.cuda.current_stream(logits.device).wait_stream(ctx.fin_stream)
        if all(g is None for g in parts):
            up = gt.float().contiguous()             # one value for the four terms (flag bit 3): no expand / copy launch
            flags |= 8
        else:
            z = torch.zeros((), dtype=torch.float32, device=logits.device)
            up = torch.stack([(g.float() if g is not None else z) for g in parts])
            if gt is not None:
                up = up + gt.float()
        dlogits = torch.empty_like(logits)
        call('stj_loss_bwd', _p(logits), _p(gt_obs), _p(gt_occ), _p(gt_flow), _p(origin), _p(coef), _p(up), _p(dlogits), B, H, W,
             flags, _st())
        return (dlogits,) + (None,) * 13


LOSS_FUSED_BWD = True      # OGMFlow_loss with an announced unit gradient (loss_fn.unit_grad, set by GraphedTrainStep) + prepare(): stj_loss_fwd_bwd
LOSS_FIN_SIDE = False      # ... with the launch that writes the loss VALUES on the caller's side stream (loss_fn.finalize_stream) instead of in front of the first backward kernel: 1355 / 1360 / 1357 against 1379 / 1380 / 1380 scenes/s on the main stream (a fork / join more in the replayed graph), profiles/r06_u_loss_one_pass.txt
LOSS_FUSED_STATS = {'hits': 0, 'misses': 0}


def loss_coef(gt_flow, gate, ogm_w, occ_w, fow, replica, flags):
    """The backward coefficients of the loss from the ground truth alone -> f32[32] (stj_loss_coef; what stj_loss_fwd writes as `coef`)."""
    _req_cuda(gt_flow)
    B, _, H, W, _ = gt_flow.shape
    cnt = zeros_f32(8, gt_flow.device).view(torch.int32)
    coef = torch.empty(32, dtype=torch.float32, device=gt_flow.device)
    call('stj_loss_coef', _p(gt_flow), _p(gate), _p(cnt), _p(coef), B, H, W, float(ogm_w), float(occ_w), float(fow), float(replica),
         int(flags) & 7, _st())
    return coef


def auc_gate_coef(gt_obs, gt_occ, gt_flow, origin, ogm_w, occ_w, fow, replica, flags):
    """auc_gate + loss_coef on one pass over the ground truth (stj_loss_gate_coef) -> (gate f32[8], coef f32[32])."""
    _req_cuda(gt_obs)
    B, _, H, W, _ = gt_obs.shape
    dev = gt_obs.device
    hist = zeros_f32(8 * 202 + 8, dev).view(torch.int32)
    gate = torch.empty(8, dtype=torch.float32, device=dev)
    coef = torch.empty(32, dtype=torch.float32, device=dev)
    call('stj_loss_gate_coef', _p(gt_obs), _p(gt_occ), _p(gt_flow), _p(origin), _p(hist), _p(gate), None, _p(coef), B, H, W,
         float(ogm_w), float(occ_w), float(fow), float(replica), int(flags) & 7, _st())
    return gate, coef


def auc_gate(gt_obs, gt_occ, gt_flow, origin, return_auc=False):
    """res_k of loss.py:127-137 for the 8 waypoints -> f32[8] (no grad)."""
    _req_cuda(gt_obs)
    B, _, H, W, _ = gt_obs.shape
    dev = gt_obs.device
    hist = zeros_f32(8 * 202, dev).view(torch.int32)
    gate = torch.empty(8, dtype=torch.float32, device=dev)
    auc = torch.empty(8, dtype=torch.float32, device=dev)
    call('stj_loss_auc_gate', _p(gt_obs), _p(gt_occ), _p(gt_flow), _p(origin), _p(hist), _p(gate), _p(auc), B, H, W, _st())
    return (gate, auc) if return_auc else gate


def ogm_flow_loss(logits, gt_obs, gt_occ, gt_flow, origin, gate, ogm_w, occ_w, fow, replica, use_warp, coef=None, unit=None, fin_stream=None):
    """coef + unit: the prepared backward coefficients (loss_coef) and the unit gradient tensor `total` will be differentiated with;
    fin_stream: a side stream for the launch that turns the sums into the loss values (the caller waits for it before reading them)."""
    return _OgmFlowLoss.apply(logits, gt_obs, gt_occ, gt_flow, origin, gate, ogm_w, occ_w, fow, replica, use_warp, coef, unit, fin_stream)
