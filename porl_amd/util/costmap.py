"""state2costmap — drop-in for /root/reference/util/costmap.py:7-64 on the device.

(b, 362) = 360 lidar ranges + relative goal (x, y)  ->  (b, 3, 360, 256) polar occupancy image: channel 0 the
one-hot beam returns (rolled by 180 rows, bin 0 cleared) plus the goal cross, channels 1-2 the goal cross.
Like the reference it zeroes entries > 8 of `state` IN PLACE.  One hand-written kernel writes the image in
channel-major order; nothing is materialised in between.  Differences: the result is contiguous (the reference
returns a permuted view), and ranges that would index past the last distance bin are dropped instead of raising.
"""
from __future__ import annotations

import torch

from .. import _native as N


def state2costmap(state, angle_bins=360, dist_bins=256):
    if state.dim() != 2 or state.shape[1] != angle_bins + 2:
        raise RuntimeError(f"state: expected (b, {angle_bins + 2}), got {tuple(state.shape)}")
    if state.dtype != torch.float32 or state.stride(1) != 1:
        raise RuntimeError("state must be fp32 with unit column stride (it is modified in place)")
    if state.device.type != "cuda":
        raise N.NativeError("state2costmap runs on a HIP device only (no CPU path)")
    b = state.shape[0]
    out = torch.empty(b, 3, angle_bins, dist_bins, dtype=torch.float32, device=state.device)
    N.check(N.lib().porl_state2costmap(N.ptr(state), state.stride(0), b, angle_bins, dist_bins, N.ptr(out),
                                       N.current_stream_ptr(state)), "porl_state2costmap")
    return out
