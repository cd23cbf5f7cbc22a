"""Occupancy / flow evaluation metrics on the GPU -- the call surface of the reference's occu_metric.py.

    compute_occupancy_flow_metrics(config, true_waypoints, pred_waypoints, no_warp=False) -> OccupancyFlowMetrics
    apply_sigmoid_to_occupancy_logits(pred_waypoint_logits) -> WaypointGrids                     # train.py:142-154

The reference evaluates these after EVERY train and validation step (train.py:243-249,280-282) with ~60 TF ops and 24
Keras AUC objects; here one HIP pass over the packed [B,H,W,32] prediction and the [B,8,H,W,*] ground truth accumulates the
histograms and sums (csrc/loss.hip, stj_metrics).  WaypointGrids built by get_pred_waypoint_logits / warpped_gt carry the
packed tensors; hand-built lists are packed with torch.cat / torch.stack first.
"""
import torch

from .loss import WaypointGrids
from .ops import _p, _st, call

FIELDS = ('vehicles_observed_auc', 'vehicles_occluded_auc', 'vehicles_observed_iou', 'vehicles_occluded_iou',
          'vehicles_flow_epe', 'vehicles_flow_warped_occupancy_auc', 'vehicles_flow_warped_occupancy_iou')


class OccupancyFlowMetrics:
    """Stand-in for occupancy_flow_metrics_pb2.OccupancyFlowMetrics: the 7 float fields set at occu_metric.py:130-139.
    `values` is the device tensor they were read from (no host sync until a field is accessed)."""

    def __init__(self, values, no_warp):
        self.values, self._host, self._no_warp = values, None, no_warp

    def __getattr__(self, name):
        if name in FIELDS:
            if self._host is None:
                self._host = self.values.tolist()
            i = FIELDS.index(name)
            return 0.0 if (self._no_warp and i >= 5) else self._host[i]
        raise AttributeError(name)


def apply_sigmoid_to_occupancy_logits(pred_waypoint_logits):
    """train.py:142-154.  The result remembers the packed logits, so the metric kernel applies the sigmoid itself."""
    g = WaypointGrids()
    v = pred_waypoint_logits.vehicles
    g.vehicles.observed_occupancy = [torch.sigmoid(x) for x in v.observed_occupancy]
    g.vehicles.occluded_occupancy = [torch.sigmoid(x) for x in v.occluded_occupancy]
    g.vehicles.flow = v.flow
    g._packed_logits = getattr(pred_waypoint_logits, '_packed', None)
    return g


def compute_occupancy_flow_metrics(config, true_waypoints, pred_waypoints, no_warp=False):
    n = config.num_waypoints
    if n != 8:
        raise NotImplementedError('num_waypoints must be 8')
    pv, tv = pred_waypoints.vehicles, true_waypoints.vehicles
    if len(pv.observed_occupancy) != n or len(tv.observed_occupancy) != n:
        raise ValueError('expected 8 waypoints in both grids')
    pred, is_logits = getattr(pred_waypoints, '_packed_logits', None), 1
    if pred is None:
        pred, is_logits = getattr(pred_waypoints, '_packed', None), 0
    if pred is None:
        pred = torch.cat([torch.cat([pv.observed_occupancy[k], pv.occluded_occupancy[k], pv.flow[k]], -1) for k in range(n)], -1)
    packed = getattr(true_waypoints, '_packed', None)
    if packed is None:
        packed = (torch.stack(tv.observed_occupancy, 1), torch.stack(tv.occluded_occupancy, 1),
                  torch.stack(tv.flow, 1), torch.stack(tv.flow_origin_occupancy, 1))
    pred = pred.detach().float().contiguous()
    if not pred.is_cuda:
        raise RuntimeError('metrics: CUDA (ROCm) tensors only: the HIP path has no CPU fallback')
    gt_obs, gt_occ, gt_flow, origin = (t.detach().float().contiguous() for t in packed)
    B, H, W, C = pred.shape
    if (H, W) != (config.grid_height_cells, config.grid_width_cells) or C != 32:
        raise ValueError(f'prediction must be [B,{config.grid_height_cells},{config.grid_width_cells},32]')
    if tuple(gt_obs.shape) != (B, 8, H, W, 1) or tuple(gt_flow.shape) != (B, 8, H, W, 2):
        raise ValueError('ground truth must be [B,8,H,W,{1,1,2,1}]')
    dev = pred.device
    hist = torch.zeros(8 * 3 * 202, dtype=torch.int32, device=dev)
    sums = torch.zeros(8 * 11, dtype=torch.float32, device=dev)
    auc = torch.empty(24, dtype=torch.float32, device=dev)
    out = torch.empty(7, dtype=torch.float32, device=dev)
    call('stj_metrics', _p(pred), _p(gt_obs), _p(gt_occ), _p(gt_flow), _p(origin), _p(hist), _p(sums), _p(auc), _p(out),
         B, H, W, is_logits, 0 if no_warp else 1, _st())
    return OccupancyFlowMetrics(out, no_warp)
