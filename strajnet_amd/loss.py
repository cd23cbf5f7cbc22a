"""OGMFlow_loss -- same constructor / call signature as the reference (loss.py:22-57), HIP-fused inside.

    OGMFlow_loss(config, ogm_weight=1000., occ_weight=1000., flow_weight=1., replica=1., flow_origin_weight=1000.,
                 no_use_warp=False, use_pred=False, use_focal_loss=True, use_gt=False)
    loss_fn(pred_waypoint_logits, true_waypoints, curr_ogm) -> {'observed_xe','occluded_xe','flow','flow_warp_xe'}

`pred_waypoint_logits` / `true_waypoints` are WaypointGrids-like objects (.vehicles.observed_occupancy /
.occluded_occupancy / .flow / .flow_origin_occupancy = lists of 8 tensors [B,H,W,{1,1,2,1}], reference
train.py:105-140).  The fast path is the packed form: `get_pred_waypoint_logits(model_out)` and
`warpped_gt(gt_obs, gt_occ, gt_flow, origin_flow)` below build WaypointGrids that remember the packed tensors they
were sliced from, so the fused kernel reads [B,H,W,32] / [B,8,H,W,*] directly; hand-built lists are packed with
torch.cat/stack first (autograd carries the gradient back through that packing).

Every flag of the reference constructor is built (template variants of the two loss kernels, csrc/loss.hip): the
train.py:195-196 configuration (use_focal_loss=False, use_gt=True, use_pred=False) and the constructor defaults
(use_focal_loss=True: tfa SigmoidFocalCrossEntropy alpha .25 gamma 2 + Keras BCE on probabilities; use_pred; no_use_warp).
"""
import torch

from . import ops

NUM_PRED_CHANNELS = 4


class _Vehicles:
    def __init__(self):
        self.observed_occupancy = []
        self.occluded_occupancy = []
        self.flow = []
        self.flow_origin_occupancy = []


class WaypointGrids:
    """Stand-in for waymo_open_dataset.utils.occupancy_flow_grids.WaypointGrids (a plain container)."""
    def __init__(self):
        self.vehicles = _Vehicles()
        self._packed = None


class OccupancyFlowTaskConfig:
    """The three fields of the Waymo proto the hot path reads (loss.py:81-82,95)."""
    def __init__(self, grid_height_cells=256, grid_width_cells=256, num_waypoints=8):
        self.grid_height_cells = grid_height_cells
        self.grid_width_cells = grid_width_cells
        self.num_waypoints = num_waypoints


def get_pred_waypoint_logits(model_outputs, num_waypoints=8):
    """train.py:105-123: slice [B,H,W,32] into per-waypoint obs/occ/flow views."""
    g = WaypointGrids()
    for k in range(num_waypoints):
        c = k * NUM_PRED_CHANNELS
        g.vehicles.observed_occupancy.append(model_outputs[..., c:c + 1])
        g.vehicles.occluded_occupancy.append(model_outputs[..., c + 1:c + 2])
        g.vehicles.flow.append(model_outputs[..., c + 2:c + 4])
    g._packed = model_outputs
    return g


def warpped_gt(gt_ogm, gt_occ, gt_flow, origin_flow):
    """train.py:126-140: GT [B,8,H,W,*] sliced along axis 1."""
    g = WaypointGrids()
    for k in range(gt_ogm.shape[1]):
        g.vehicles.observed_occupancy.append(gt_ogm[:, k])
        g.vehicles.occluded_occupancy.append(gt_occ[:, k])
        g.vehicles.flow.append(gt_flow[:, k])
        g.vehicles.flow_origin_occupancy.append(origin_flow[:, k])
    g._packed = (gt_ogm, gt_occ, gt_flow, origin_flow)
    return g


class LossDict(dict):
    """The reference's loss dict (loss.py:161-170) plus `.total`: the sum of the four entries (train.py:221 `tf.add_n`) as ONE
    tensor produced next to them -- differentiating `.total` instead of `sum(d.values())` saves a dozen tiny launches."""
    total = None
    packed = None


class OGMFlow_loss:
    def __init__(self, config, ogm_weight=1000.0, occ_weight=1000.0, flow_weight=1.0, replica=1.0,
                 flow_origin_weight=1000.0, no_use_warp=False, use_pred=False, use_focal_loss=True, use_gt=False):
        self.config = config
        self.use_focal_loss, self.use_pred = bool(use_focal_loss), bool(use_pred)
        self.ogm_weight, self.occ_weight = ogm_weight, occ_weight
        self.flow_weight = flow_weight            # stored but never applied, as in the reference (loss.py:29,141)
        self.replica = replica
        self.flow_origin_weight = flow_origin_weight
        self.no_use_warp, self.use_gt = no_use_warp, use_gt
        # Not in the reference: a caller that will differentiate `.total` with a known all-ones gradient tensor (GraphedTrainStep:
        # `total.backward(one)`) announces that tensor here; together with prepare() the loss then writes d(total)/d(logits) in its
        # forward pass (stj_loss_fwd_bwd) and backward hands it over when it is called with exactly that tensor -- any other
        # upstream gradient takes the general kernel.
        self.unit_grad = None
        self.finalize_stream = None     # with unit_grad: a side stream for the launch that writes the loss VALUES (the caller joins it)

    def _flags(self):
        return (0 if self.no_use_warp else 1) | (2 if self.use_focal_loss else 0) | (4 if self.use_pred else 0)

    @staticmethod
    def _ground_truth(true_waypoints):
        tv = true_waypoints.vehicles
        packed = getattr(true_waypoints, '_packed', None)
        if packed is None:
            packed = (torch.stack(tv.observed_occupancy, 1), torch.stack(tv.occluded_occupancy, 1),
                      torch.stack(tv.flow, 1), torch.stack(tv.flow_origin_occupancy, 1))
        return tuple(t.float().contiguous() for t in packed)

    def prepare(self, true_waypoints):
        """Optional: compute what depends on the ground truth alone (the Keras-AUC gate of the flow terms, loss.py:127-137) ahead of
        the model's forward pass, e.g. on a side stream; the next __call__ with the SAME true_waypoints object then skips it.  In
        the train step the gate (37 us) otherwise sits between the last forward and the first backward kernel, where nothing overlaps it."""
        self._prepared = None
        if self.use_gt or self.unit_grad is not None:
            gt = self._ground_truth(true_waypoints)
            w = (self.ogm_weight, self.occ_weight, self.flow_origin_weight, self.replica, self._flags())
            coef = None
            if self.use_gt and self.unit_grad is not None:      # the backward coefficients depend on the ground truth alone as well:
                gate, coef = ops.auc_gate_coef(*gt, *w)         # the gate's histogram pass counts the flow term's denominator on the way
            else:
                gate = ops.auc_gate(*gt) if self.use_gt else torch.ones(8, dtype=torch.float32, device=gt[0].device)
                if self.unit_grad is not None:
                    coef = ops.loss_coef(gt[2], gate, *w)
            self._prepared = (true_waypoints, gate, coef)

    def __call__(self, pred_waypoint_logits, true_waypoints, curr_ogm=None):
        n = self.config.num_waypoints
        if n != 8:
            raise NotImplementedError('num_waypoints must be 8')
        pv, tv = pred_waypoint_logits.vehicles, true_waypoints.vehicles
        if len(pv.observed_occupancy) != n or len(tv.observed_occupancy) != n:
            raise ValueError('expected 8 waypoints in both grids')
        logits = getattr(pred_waypoint_logits, '_packed', None)
        if logits is None:
            logits = torch.cat([torch.cat([pv.observed_occupancy[k], pv.occluded_occupancy[k], pv.flow[k]], -1)
                                for k in range(n)], -1)
        gt_obs, gt_occ, gt_flow, origin = self._ground_truth(true_waypoints)
        B, H, W, Cc = logits.shape
        if (H, W) != (self.config.grid_height_cells, self.config.grid_width_cells) or Cc != 32:
            raise ValueError(f'logits must be [B,{self.config.grid_height_cells},{self.config.grid_width_cells},32]')
        if tuple(gt_obs.shape) != (B, 8, H, W, 1) or tuple(gt_flow.shape) != (B, 8, H, W, 2):
            raise ValueError('ground truth must be [B,8,H,W,{1,1,2,1}]')
        cached, self._prepared = getattr(self, '_prepared', None), None
        coef = None
        if cached is not None and cached[0] is true_waypoints:
            gate, coef = cached[1], cached[2]            # computed by prepare() (they depend on the ground truth only)
        elif self.use_gt:
            gate = ops.auc_gate(gt_obs, gt_occ, gt_flow, origin)
        else:
            gate = torch.ones(8, dtype=torch.float32, device=logits.device)
        loss = ops.ogm_flow_loss(logits, gt_obs, gt_occ, gt_flow, origin, gate, self.ogm_weight, self.occ_weight,
                                 self.flow_origin_weight, self.replica, self._flags(),
                                 coef=coef, unit=self.unit_grad if coef is not None else None, fin_stream=self.finalize_stream)
        d = LossDict({'observed_xe': loss[0], 'occluded_xe': loss[1], 'flow': loss[2],
                      'flow_warp_xe': loss[3] if not self.no_use_warp else 0.0})
        d.total, d.packed = loss[4], loss[5]       # the sum (differentiable) and the 4 values as one detached vector
        return d
