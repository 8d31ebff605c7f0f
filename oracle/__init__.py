"""
TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the py4cast hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package.  The product (``py4cast_amd``) never does: it fails loudly
when the HIP extension is missing.

Parity status
-------------
* ``oracle.rollout`` / ``oracle.losses``: PINNED.  Checked bit-for-bit / <=1e-6 against
  golden vectors produced by executing the *unmodified* reference files
  ``py4cast/lightning.py`` and ``py4cast/losses.py`` (stub-imported, see
  ``tests/golden/make_golden.py``); fixtures live in ``tests/golden/*.npz``.
* ``oracle.halfunet``: PARITY UNPINNED.  The network arithmetic lives in the
  third-party package ``mfai`` v5.0.1 (``requirements.txt:26`` of the reference),
  which is absent from ``/root/reference`` and not installable here.  The oracle
  restates the published Half-UNet architecture as used by mfai from plain
  ``torch.nn.functional`` ops; the HIP kernels are compared against that.
* ``oracle.graph`` / ``oracle.graphlam`` (mesh GNN: index_select + cat + index_add_) and
  ``oracle.window_attention`` / ``oracle.swinunetr`` (Swin: roll, window_partition, softmax
  attention with relative position bias and shift mask): PARITY UNPINNED for the same reason
  (GraphLam and SwinUNetR are mfai classes); each restatement is cross-checked on CPU against an
  independent formulation of the published operation (tests/test_widen_oracle_cpu.py).
"""
