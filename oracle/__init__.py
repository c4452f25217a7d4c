"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the PCG / curvature-matvec hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline -- never as the
thing that is shipped.  The product package ``pytorchhessianfree_amd`` must not
import this package (``tests/test_boundary.py`` greps for that).

Parity status: PINNED.  ``oracle.pcg`` is checked bit-for-bit against the real
reference ``hessianfree.cg.cg`` in the build container by
``tests/golden/make_golden.py`` (which imports ``/root/reference`` there) and,
everywhere else, against the committed golden vectors in ``tests/golden/``.
"""
