"""CPU oracle for the SqueezeDet hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``squeezedet_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker (never the thing measured as
the product, never shipped).

Parity status
-------------
* NumPy half of the reference (``set_anchors``, ``util.batch_iou/nms/
  bbox_transform(_inv)``, ``ModelSkeleton.filter_prediction``): **pinned** --
  ``oracle/ref_numpy_half.py`` imports the reference's own functions unchanged
  (only possible where ``/root/reference`` exists), ``tests/golden/
  make_golden.py`` ran them on seeded inputs and committed the outputs under
  ``tests/golden/``; the restatement in ``oracle/sqdet_oracle.py`` is checked
  against those vectors by ``tests/test_oracle_golden.py``.
* Label assignment (``imdb.read_batch``, dataset/imdb.py:120-260): **pinned** --
  ``oracle/ref_imdb_half.py`` imports the reference's ``dataset/imdb.py`` unchanged
  with ``cv2`` stubbed (the label half never looks at pixels), ``make_golden.py``
  ran it on seeded annotations -> ``tests/golden/labels.npz``;
  ``train_oracle.assign_anchors`` and the HIP kernel ``sqdet_build_labels`` are
  checked against it bit for bit.  Ties between free anchors are resolved by
  ``np.argsort``'s unspecified order in the reference: the cases are tie-free.
* TF-graph half (conv / bias / relu / max-pool / softmax / sigmoid / exp):
  **parity unpinned**.  The arithmetic lives in tensorflow-gpu==1.0.0
  (requirements.txt:6), which is not vendored, not installed and cannot run
  here; the reference ships no tests or golden vectors.  It is restated from
  the reference's call sites (nn_skeleton.py:471-586,142-283;
  nets/squeezeDet.py:30-106) plus TF's documented op semantics.
"""
