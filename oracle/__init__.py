"""CPU oracle for the DiffPhore reverse-diffusion hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU / numpy, fp32 or fp64) of the reference algorithm for the path named in
BASELINE.json: the score model forward (`src/models/score_model_phore.py`) and the sampler + conformer update
(`src/utils/sampling.py`, `src/utils/diffusion_utils.py`, `src/utils/torsion.py`, `src/utils/geometry.py`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and only as the checker.
Nothing under `diffphore_amd/` imports it: the product path is the HIP library and fails loudly without it.

Pinning status:
  * conformer-update half (axis-angle, Kabsch, torsion update, modify_conformer incl. the norm-slot quirk, sigma
    schedule, sinusoidal embedding, SO(3)/torus score-norm tables): PINNED against outputs of the reference's own
    functions run in the build container (`tests/golden/make_golden.py` -> `tests/golden/*.npz`,
    `diffphore_amd/data/*.npy`).
  * score-model half, reference-authored logic (effective configuration, graph construction and edge order, feature
    concatenation, cross-edge sign / normal rules, MLP and irreps wiring, heads, sigma scaling, state-dict layout) and the
    sampler loop (RNG draw order, per-step constants, per-sample conformer updates): PINNED against the reference's OWN model
    and sampler code - `src/models/score_model_phore.py` built by the reference's `get_model` from the released yml, and
    `src/utils/sampling.py::randomize_position / sampling_phore`, executed in the build container by
    `tests/golden/make_model_golden.py` -> `tests/golden/ref_model_glue.npz` (`tests/test_oracle_model_golden.py`).
  * score-model half, third-party conventions: PARITY UNPINNED.  e3nn==0.5.1 / torch_scatter / torch_cluster /
    torch_geometric are absent from /root/reference and from this image (the golden run above adapts them onto THIS package's
    restatements, so it cannot see an error in them), the reference has no tests or golden vectors, and the trained checkpoint
    is missing.  The e3nn operators are restated from their published definitions (real spherical harmonics, Wigner-3j in
    e3nn's real basis, FullyConnectedTensorProduct 'uvw' with component/element normalisation, FullTensorProduct 'uvuv',
    e3nn.nn.BatchNorm eval) and locked by property tests (equivariance, orthogonality, normalisation) in
    `tests/test_oracle_e3.py`; torch_cluster.radius(_graph) follows its CUDA kernel's documented semantics.
"""
