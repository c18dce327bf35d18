"""CPU ORACLE for the V2X GNN Q-network hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in the product path (the `globecom2020-resourceallocationgnn_amd/` package, the
C-ABI library) may import, call, link or execute anything under `oracle/`.  Allowed users:
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`, and there only
as the checker / the CPU number printed beside the GPU number.

What it restates: the Keras/TF1 graph built by `/root/reference/BS_brain.py:108-216`
(GNNLayer `:17-56`, AggLayer `:60-82`, huber `:86-87`, Adam `:212`) and the fit/predict
semantics of `:218-235`.

PARITY STATUS: the reference's arithmetic lives in Keras 2.2.4 / TensorFlow 1.14, which are
neither vendored in the reference tree nor installable here, and the reference ships no
tests, golden vectors or trained weights.  The oracle is therefore pinned by
  (1) golden vectors produced by executing the reference's OWN `_create_model`,
      `GNNLayer.call` and `AggLayer.call` code (imported from /root/reference in the build
      container) against a numpy-eager stand-in for the Keras *library* calls
      (`tests/golden/make_golden.py`, fixtures in `tests/golden/*.npz`);
  (2) torch-autograd (float64) and finite-difference checks of the hand-written backward;
  (3) equality of the three formulations (literal dense-kron / compact dense / CSR).
With respect to a real Keras/TF1 run it remains **parity unpinned** (see DESIGN.md).
"""
from .spec import GnnSpec  # noqa: F401
