"""oracle/ -- CPU restatement of the reference's algorithm for the NAF hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under neuralvolumetricreconstructionformedicalimages_amd/ imports this
package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker
(or, for cpu_baseline, as the reported CPU "port" baseline) -- never as the thing shipped or measured as
the product.

Pinning:
  * render / MLP / geometry / loss / metrics restatements are pinned against golden vectors captured by
    importing the reference's own pure-PyTorch modules in the build container
    (tests/golden/make_golden.py -> tests/golden/*.npz).
  * the hash-grid encoder has no runnable reference here (CUDA only, does not build under ROCm) and the
    reference holds no tests or golden vectors for it: "parity unpinned" by the reference; pinned by
    hand-derived integer KATs, analytic properties, and two independently written restatements
    (hash_ref.c scalar C, hashgrid_ref.py vectorised numpy) that must agree.
"""
