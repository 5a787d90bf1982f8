"""CPU oracle for the da-sac hot path -- TEST INFRASTRUCTURE ONLY.

A CPU restatement (PyTorch fp32 tensor ops, closed forms) of the reference's per-step
hot path (`/root/reference/models/{sac,deeplabv2,fcn,basenet}.py`, step order of
`/root/reference/train.py:119-155,211-250`).  Every function cites the reference
file:line it follows.

Rules (enforced by tests/test_models_cpu.py::test_product_never_imports_the_oracle):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
    import this package -- and only as the checker / the timed CPU baseline;
  * the product (`da-sac_amd/`) never imports it and has no CPU fallback.

Pinning: the reference ships no tests or golden vectors, so the oracle
is pinned against outputs of the reference itself, imported on CPU in the build container
by `tests/golden/make_goldens.py`; the captured vectors live in `tests/golden/*.npz` and
`tests/test_oracle_golden.py` re-checks the oracle against them everywhere.
Third-party arithmetic underneath the reference is ATen (torch 2.10.0 CPU kernels):
`upsample_bilinear2d`, `affine_grid`/`grid_sampler_2d`, `softmax`, `cross_entropy`,
`conv2d`, `batch_norm`, `max_pool2d`; the interpolation / warping / loss formulas are
restated explicitly in `head_ref.py`, conv/BN/pool are delegated to ATen (and checked against
numpy loops in tests/test_oracle_nets.py); and Pillow 12.2.0 for the target views
(`Image.resize` BILINEAR / NEAREST, restated in `views_ref.py`, pinned against Pillow
itself and golden g12).  `step_ref.ThreadWorld` restates what DistributedDataParallel and the
reference's two all_gathers do for R ranks (pinned by golden g11).
"""
