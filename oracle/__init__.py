"""CPU oracle for the Data-Driven MPC per-timestep QP path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``direct_data_driven_mpc_amd`` (the
product) may import this package: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.
"""
