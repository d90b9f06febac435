"""oracle/ — CPU restatement of the vision-ft hot path.  TEST INFRASTRUCTURE ONLY.

Plain PyTorch-CPU / numpy restatements of the reference algorithms on the denoiser
forward/backward path, each function citing the reference file:line it follows.
The restatement is pinned against the reference itself: ``oracle/gen_golden.py`` imports the
reference's Python from /root/reference (possible in the build container only), runs it on seeded
inputs and stores inputs + outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks
every oracle function against those fixtures.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package — as the checker, never as the thing measured or shipped.  ``vision_ft_amd`` never
imports it and has no CPU fallback.
"""
