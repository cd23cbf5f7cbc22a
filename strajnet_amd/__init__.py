"""strajnet_amd -- MI355X-native (gfx950 / CDNA4) implementation of the STrajNet forward/backward hot path.

Drop-in call surface of the reference's two classes:
    from strajnet_amd import STrajNet, OGMFlow_loss
The math runs in hand-written HIP kernels (strajnet_amd/csrc -> libstrajnet_hip.so, C ABI in include/strajnet_hip.h).
"""
from .modules import STrajNet            # noqa: F401
from .loss import (OGMFlow_loss, WaypointGrids, OccupancyFlowTaskConfig,     # noqa: F401
                   get_pred_waypoint_logits, warpped_gt)
from .optim import Nadam                # noqa: F401
from .metrics import compute_occupancy_flow_metrics, apply_sigmoid_to_occupancy_logits, OccupancyFlowMetrics     # noqa: F401
