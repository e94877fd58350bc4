"""CPU ORACLE -- test infrastructure, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product package ``pointcloudmatters_amd`` never does (and fails loudly when its
HIP library is missing instead of falling back to anything here).

Contents
--------
* ``pcm_oracle.c/.h``  -- plain-C restatement of the reference's CUDA kernels K1-K9
  (``/root/reference/libs/pointops/src/**``), un-contracted fp32.  Parity against a *running*
  reference is unpinned (the reference is CUDA-only and has no golden vectors); see the header.
* ``py_twin.py``        -- an independent pure-Python/NumPy twin of K1-K4 for small cases; the
  tests require the two restatements to agree bit-for-bit.
* ``pointops_cpu.py``   -- the reference's ``pointops`` Python API (``libs/pointops/functions``)
  on CPU torch tensors, backed by the C oracle.
* ``sa_reference.py``   -- the reference's set-abstraction layer / PointNet / ACT step composed in
  plain PyTorch (CPU) in the reference's op order -- the "port" timed as ``cpu_baseline``.
"""
