"""CPU oracle for the candidate-evaluation hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy float64 for the schedule tables,
PyTorch-CPU float32 functional ops for the networks) of the reference
algorithm for the path named in BASELINE.json ``north_star``:
``ddim_sample_loop`` / ``p_sample_loop`` -> ``UNetModel.forward`` -> uint8
image batch -> FID statistics.  Every function cites the reference file:line
it restates.

Pinning: the oracle is checked against golden vectors captured by importing
the reference's own ``guided_diffusion`` modules in the build container
(``tests/golden/capture_golden.py`` -> ``tests/golden/*.npz``; see
``tests/test_oracle_golden.py``).  The Inception-v3 feature extractor is the
one piece with no reference fixture ("parity unpinned").

Rules (enforced by ``tests/test_layout.py``):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import this package;
  * nothing under ``autodiffusion_amd/`` imports it -- the product path is the
    HIP extension and fails loudly when that extension is missing.
"""
