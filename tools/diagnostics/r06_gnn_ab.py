"""bench.py with the round-5 mesh-GNN routes restored (A/B reference for csrc/nodeproj.hip): P4C_R06_OLD_PROJ=1 -> the node projections as
separate row-GEMM launches (ops_rows.row_linear_multi), P4C_R06_NO_DEFER=1 -> every gradient partial reduced by its own launch.
P4C_R06_NO_BATCH_PREP=1 -> ops_gemm.BATCHED_PREP = False.  Same flags as bench.py; its lines are never the judged ones."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from py4cast_amd import ops_nodeproj as NP  # noqa: E402
from py4cast_amd import ops_rows as R  # noqa: E402

if os.environ.get("P4C_R06_NO_DEFER") == "1":
    NP.GradQueue.enabled = False
if os.environ.get("P4C_R06_OLD_PROJ") == "1":
    def old(x, weights, grads_in_place=True):
        weights = list(weights)
        if len(weights) == 1:
            return (R.row_linear(x, weights[0], grads_in_place=grads_in_place),)
        return R.row_linear_multi(x, weights, grads_in_place=grads_in_place)
    NP.node_proj = old
if os.environ.get("P4C_R06_NO_BATCH_PREP") == "1":      # one weight-image preparation launch per weight (before the batched form)
    from py4cast_amd import ops_gemm as G  # noqa: E402

    G.BATCHED_PREP = False
bench.main()
