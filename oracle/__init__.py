"""CPU oracle for the MMPL chunk-AR denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mmpl_amd/`` may import this package; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and there
only as the checker / the reported CPU baseline -- never as the thing shipped or measured.

The oracle is a from-scratch restatement (plain PyTorch CPU ops, reference rounding points) of
the algorithms in Tele-AI/MMPL's ``MMPL_t2v`` tree; every function cites the reference
file:line it follows.  It is *pinned*: ``tests/golden/make_golden.py`` imports the real
reference (in the build container only) on seeded synthetic weights and commits its outputs
as fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against those fixtures.  Real-checkpoint parity is unpinned (no weights in the container).
"""
