"""CPU oracle for the bot7 GP-posterior + acquisition hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``bot7_amd``) never does and fails loudly when ``libbot7hip.so`` is missing.

Layout
  b7_oracle.c  plain-C restatement of the integer / operation-order-exact pieces (Sobol, erf, EI, CB,
               marginalisation, argmax, row removal, pdist, jitter Cholesky), one rounded op per Lua op.
  cport.py     ctypes binding of the above.
  gp.py        numpy/scipy (BLAS dgemm, LAPACK dpotrf/dtrtrs -- the library class Torch7 calls) GP
               regression in Cholesky form.  PARITY UNPINNED: gp.models.gp_regressor is not in the
               reference tree (models/init.lua:15) and the reference holds no fixtures for it.
"""
