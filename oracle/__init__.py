"""CPU oracle for the rltime Q-learning hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product (``rltime_amd``) never imports this package and fails loudly when
its HIP library is missing.

What it is: a from-scratch restatement (plain Python + numpy, torch-CPU fp32 for
the floating-point target/loss math) of the reference algorithm for the path

    History.update -> (uniform | prioritized sum-tree) sampling ->
    n-step / sequence batch assembly with stored recurrent state ->
    value-rescaled double-Q / IQN targets -> DQN / IQN loss ->
    update_losses (sequence priorities)

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Parity status: PINNED — the restatement is checked against
golden vectors produced by importing the *unmodified* reference in the
development container (``tests/golden/generate.py``; the reference ships no
tests or fixtures of its own), and the generator itself asserts
oracle == reference while producing them.

Scalar-type fidelity matters here: the reference keeps Python floats,
``np.float32`` and ``np.float64`` scalars in its tree and in its n-step
returns, so NumPy-2 promotion rules decide the rounding of every add
(SURVEY.md Appendix A-6).  The oracle therefore deliberately computes with the
same scalar objects instead of arrays.
"""
