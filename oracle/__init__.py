"""ORACLE -- test infrastructure, not product code.

CPU restatement (plain PyTorch fp32 + autograd) of the reference's hot path
(``scripts/learned_multi_view_recon_nn.py`` -> ``nemo.neural_motion_model.NemoV*.step``).
Parity status: PINNED -- checked function-by-function and trajectory-by-trajectory
against golden vectors recorded from the real reference in the build container
(``tools/gen_golden.py`` -> ``tests/golden/*.npz``; checker: ``tests/test_oracle_golden.py``).

Allowed importers: ``tests/``, ``__graft_entry__.smoke()``, ``bench.py`` (``cpu_baseline`` leg).
The product package ``nemo_cvpr2023_amd`` must never import this package.
"""
