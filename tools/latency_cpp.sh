#!/bin/bash
# Per-frame wall time of the C++ host mirror's DisparityCb (host/d2pc_replay latency) at the reference's
# native 752x480 mono16 input: pageable PointCloud2 payload (as the reference's message) against the
# pinned-allocator payload the kernels store into directly.  GPU box only.
cd "$(dirname "$0")/.."
python - <<'PY'
import sys
sys.path.insert(0, '.')
from disparity_to_point_cloud_amd.synth import synth_disparity
synth_disparity(1, 0, 752, 480, "mono16").tofile("gpurun_out/in752.raw")
PY
for p in "" pinned; do for m in "" compact; do host/d2pc_replay latency gpurun_out/in752.raw 752 480 mono16 300 $m $p; done; done
