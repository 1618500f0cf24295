"""CPU oracle for the ClavaDDPM hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain torch-CPU / numpy
restatement of the reference algorithm (file:line cited per function).  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import it; the shipped package `clavaddpm_amd` never does, and fails loudly
when its HIP library is missing.

Parity pin: every function here is checked against golden vectors captured by
running the reference itself in the build container
(`tests/golden/make_golden.py`, fixtures under `tests/golden/*.npz`).  The GMM
E-step (scikit-learn, third-party, pinned upstream at 1.4.1.post1; 1.7.2 in
this image) is pinned only by vectors captured from the installed 1.7.2 - the
reference holds no fixture of its own for it.
"""
from . import schedule, gaussian, multinomial, network, guidance, chain, loss, gmm, trainer  # noqa: F401
