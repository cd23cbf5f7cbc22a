"""hipGraph capture of the whole train step (forward + OGMFlow_loss + backward into the flat gradient bucket).

The step is ~650 stream-ordered kernel launches with static shapes, no host synchronisation and no allocation outside
torch's graph-private pool (every kernel of libstrajnet_hip.so launches on the current stream and never allocates), so
it is capturable as ONE graph; replaying it removes the Python / launch overhead (~10 ms per step, measured with B=1).
New data is fed by copying into the static input tensors; weights are read in place (the bf16 shadow cast is part of
the graph), gradients land in model.flat_grads().  The data-parallel all-reduce stays outside the graph.
The Dropout / DropPath stream advances inside the graph (stj_rng_advance bumps a device counter), so every replay draws new masks.
"""
import torch

from . import ops
from .loss import get_pred_waypoint_logits, warpped_gt


class GraphedTrainStep:
    """split=False: ONE graph for the whole step.  split=True (data parallel): TWO graphs cut at the raster encoder's outputs
    (model.cut_encoder) -- graph A = forward + loss + backward of everything downstream of the encoder, graph B = the encoder's
    backward; `__call__(between=f)` runs f between the two replays (dp.OverlappedGradSync.tail: the tail bucket's all-reduce then
    overlaps graph B).  Both graphs share one private memory pool: B reads the activations A saved."""

    def __init__(self, model, loss_fn, batch, warmup=2, training=True, split=False, keep_graph=False):
        """keep_graph: leave the captured hipGraph_t un-instantiated (torch.cuda.CUDAGraph(keep_graph=True)) so that a caller can edit it
        (self.graph.raw_cuda_graph(): e.g. kernel-node attributes, tools/probes/graph_node_priority.py) before self.graph.instantiate()."""
        self.model, self.loss_fn, self.training, self.split = model, loss_fn, training, split
        self.static = {k: v.clone() for k, v in batch.items()}
        self.graph = self.graph_b = None
        self._side = None
        self.losses = None
        # the gradient `total` is differentiated with: created up front and announced to the loss, which then writes d(total)/d(logits)
        # in its forward pass (OGMFlow_loss.unit_grad, ops.LOSS_FUSED_BWD)
        self._one = torch.ones((), dtype=torch.float32, device=next(iter(batch.values())).device)
        if hasattr(loss_fn, 'unit_grad'):
            loss_fn.unit_grad = self._one
        if split:
            model.cut_encoder = True
        side = ops.role_stream(torch.cuda.current_device(), 'warmup')
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager()
                if split:
                    model.backward_encoder()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(keep_graph=True) if keep_graph else torch.cuda.CUDAGraph()
        # thread_local: calls that are illegal during capture only count against THIS thread -- a process-group watchdog thread
        # (RCCL, data parallel runs) querying its events must not invalidate the capture
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            self.losses = self._eager()
        self.graph = g
        if split:
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=g.pool(), capture_error_mode='thread_local'):
                model.backward_encoder()
            self.graph_b = gb

    def _eager(self):
        x, m = self.static, self.model
        m.zero_grad()
        tw = warpped_gt(x['gt_obs'], x['gt_occ'], x['gt_flow'], x['origin_flow'])
        main = torch.cuda.current_stream()
        if hasattr(self.loss_fn, 'prepare'):         # the ground-truth-only part of the loss: side stream, under the forward pass --
            if self._side is None:                   # issued from the model's mid-forward hook (behind the encoder's first stage: at the
                self._side = ops.role_stream(torch.cuda.current_device(), 'loss_prep')     # head of the step its two launches delayed the first encoder kernel)
                if hasattr(self.loss_fn, 'finalize_stream'):
                    self.loss_fn.finalize_stream = self._side

            def prepare():
                self._side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._side):
                    self.loss_fn.prepare(tw)
            m.mid_forward_hook = prepare
        try:
            out = m(x['ogm'], x['map_img'], training=self.training, obs=x['obs'], occ=x['occ'], mapt=x.get('mapt'), flow=x['flow'])
        finally:
            m.mid_forward_hook = None
        if self._side is not None:
            main.wait_stream(self._side)
        d = self.loss_fn(get_pred_waypoint_logits(out), tw, None)
        total = d.total                      # observed_xe + occluded_xe + flow + flow_warp_xe (train.py:221)
        total.backward(self._one)             # (an explicit tensor: backward()'s implicit ones_like is a fill launch on every replay)
        if self._side is not None and ops.LOSS_FIN_SIDE:
            main.wait_stream(self._side)      # the loss values' finalize launch (issued behind the loss pass on the side stream: off, see ops.LOSS_FIN_SIDE)
        self.total = total.detach()           # the sum the finalize kernel wrote (static across replays, like self.losses)
        return d.packed             # [observed_xe, occluded_xe, flow, flow_warp_xe], detached

    def load(self, batch):
        """Copy a new batch into the static input tensors (stream-ordered device copies)."""
        for k, v in batch.items():
            if k in self.static:
                self.static[k].copy_(v, non_blocking=True)

    def __call__(self, batch=None, between=None):
        if batch is not None:
            self.load(batch)
        self.graph.replay()
        if self.graph_b is not None:
            if between is not None:
                between()
            self.graph_b.replay()
        return self.losses          # [observed_xe, occluded_xe, flow, flow_warp_xe]; gradients in model.flat_grads()


class GraphedForward:
    """hipGraph capture of the inference forward (training=False, no autograd): BASELINE config 4 (batch 32, one MI355X).
    __call__(batch) copies the batch into the static inputs, replays, and returns the static [B,Hg,Hg,32] f32 output.

    pipeline_agents=True: the agent branch (trajNet: ~30 dependent launches of a few microseconds that a replayed graph starts only when
    the raster encoder is through -- 0.3 of the 6.3 ms B = 32 step with nothing beside them, profiles/r05_c_timeline_infer_b32_f16.txt) is
    taken OUT of the graph and launched on a second stream.  prefetch_agents(next_batch), called right after __call__(batch), runs it for the NEXT batch under
    this batch's raster path; __call__ then waits for it, copies its two small results into the main graph's static inputs and replays the
    main graph, which no longer contains the branch.  Without a prefetch the agent branch runs in front of the main graph (same result).
    (Also tried: a replay that does not re-cast / re-pack / re-fold the constant weights -- one cast + seven side-stream launches less per step --
    measured 1.3 % SLOWER in three same-box pairs, 5333 / 5118 / 5333 vs 5390 / 5206 / 5412 scenes/s: where those launches sit decides when the
    executor starts the other branches, DESIGN 4e.  Not kept.)"""

    def __init__(self, model, batch, warmup=2, pipeline_agents=False):
        self.model = model
        self.static = {k: v.clone() for k, v in batch.items() if k in ('ogm', 'map_img', 'obs', 'occ', 'flow')}
        self.pipeline_agents = pipeline_agents
        self._prefetched = False
        self._staged = None
        self._weights_seen = -1
        side = ops.role_stream(torch.cuda.current_device(), 'warmup')
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if pipeline_agents:
            # the model's own agent-branch stream (idle in a main graph captured without the branch).  NOT one more stream: with a sixth
            # side stream alive in the process a LATER, unrelated graph replay died in hip::Graph::UpdateStreams (ROCm 7.2; tests/test_model_gpu.py
            # run as a whole: bisected to the number of distinct streams in use -- see ops.role_stream)
            self.agent_stream = model._streams[0] if model._streams[0] is not None else ops.role_stream(torch.cuda.current_device(), 'side')
            self.agent_done, self.agent_free = torch.cuda.Event(), torch.cuda.Event()
            # (the branch is launched EAGERLY, ~30 small launches issued by the host while the GPU replays the main graph; as a hipGraph of its
            #  own it ran as fast)
            with torch.no_grad():
                self.agent_next = model.agent_encode(self.static['obs'], self.static['occ'])
            self.agent_cur = tuple(t.clone() for t in self.agent_next)
            torch.cuda.synchronize()
            model.agent_override = self.agent_cur
        try:
            self.graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                self.out = self._eager()
        finally:
            model.agent_override = None
        if pipeline_agents:
            self.agent_free.record(torch.cuda.current_stream())

    def _eager(self):
        x = self.static
        return self.model(x['ogm'], x['map_img'], training=False, obs=x['obs'], occ=x['occ'], mapt=None, flow=x['flow'])

    def prefetch_agents(self, batch=None, ready=None):
        """Start the agent branch of the NEXT batch (None: the static inputs as they are) on the agent stream, under whatever the main
        stream is running.  `ready`: an event recorded behind whatever produced batch['obs'] / batch['occ'] (e.g. their host-to-device
        copies on a feed stream); the agent stream waits for it.  Without one the agent stream is ordered behind everything the CALLER's
        current stream has been given so far -- always safe, but when that stream is the one replaying the main graph the branch then runs
        after the replay instead of under it: prefer `__call__(batch, next_batch=...)`, which stages the tracks in front of the replay."""
        if not self.pipeline_agents:
            return
        st = self.agent_stream
        st.wait_event(self.agent_free)               # the previous results have been copied out
        if batch is not None:
            if ready is not None:
                st.wait_event(ready)
            else:
                st.wait_stream(torch.cuda.current_stream())
        if self._staged is not None:                 # tracks staged on the main stream in front of the replay (__call__(next_batch=))
            st.wait_event(self._staged)
            self._staged = None
        with torch.cuda.stream(st):
            if batch is not None:
                for k in ('obs', 'occ'):
                    if k in batch:
                        self.static[k].copy_(batch[k], non_blocking=True)
                        batch[k].record_stream(st)
            with torch.no_grad():
                # (no weight cast here: the main graph casts the same buffer at the head of every replay, and two casts of one buffer on two
                #  streams are an unordered write / read pair)
                self.agent_next = self.model.agent_encode(self.static['obs'], self.static['occ'], cast=False)
            self.agent_done.record(st)
        self._prefetched = True
        self._weights_seen = self.model.weights_version

    def __call__(self, batch=None, next_batch=None):
        """next_batch (pipeline_agents only): the batch of the NEXT call; its agent tracks are copied into the static inputs on the
        current stream IN FRONT of this replay (ordered behind whatever produced them on this stream) and the agent branch for them
        starts on the agent stream under the replay."""
        main = torch.cuda.current_stream()
        if self.pipeline_agents and self._prefetched and self._weights_seen != self.model.weights_version:
            self._prefetched = False                 # the prefetched encoding is of the OLD weights (load_weights since): run it again
        if self.pipeline_agents and not self._prefetched:
            self.model._sync_compute_weights()       # (the branch runs in front of the replay: it needs the current compute copy now)
            self.agent_stream.wait_stream(main)
            self.prefetch_agents(batch)
        if batch is not None:
            for k, v in batch.items():
                if k in self.static and not (self.pipeline_agents and k in ('obs', 'occ')):
                    self.static[k].copy_(v, non_blocking=True)
        if self.pipeline_agents:
            main.wait_event(self.agent_done)
            for d, s_ in zip(self.agent_cur, self.agent_next):
                d.copy_(s_, non_blocking=True)
            self.agent_free.record(main)
            self._prefetched = False
            if next_batch is not None:               # the agent branch has read the static tracks (agent_done): they may be overwritten
                for k in ('obs', 'occ'):
                    if k in next_batch:
                        self.static[k].copy_(next_batch[k], non_blocking=True)
                self._staged = torch.cuda.Event()
                self._staged.record(main)
        self.graph.replay()
        if self.pipeline_agents and next_batch is not None:
            self.prefetch_agents()
        return self.out
