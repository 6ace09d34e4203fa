"""CPU oracle for the CHAMELEON NAR training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``chameleon_recsys_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and
there only as the checker / the timed CPU baseline, never as the product path.

Parity status: **parity unpinned at the TensorFlow boundary**.  The reference's graph
(``nar_module/nar/nar_model.py``) needs ``tensorflow==1.12.3`` which is not installable here and
the reference ships no golden vectors for logits / loss / gradients.  What *is* pinned:

* ``oracle.state``   - against the reference's own ``nar/clicked_items_state.py`` executed in the
  build container (fixtures in ``tests/golden/state_*.npz``, generator ``oracle/make_golden.py``);
* ``oracle.sampler`` - against the reference's 8 property tests
  (``nar/benchmarks/candidate_sampling_tests.py:10-99``) and, distribution-wise, against the
  reference's numpy clone ``nar/benchmarks/candidate_sampling.py``;
* ``oracle.metrics`` (HitRate/MRR) - against ``nar/metrics.py`` executed here (fixtures).

Everything inside the TF graph is a line-by-line restatement (each function cites the
reference lines it follows) in float32 PyTorch-CPU / numpy.
"""
