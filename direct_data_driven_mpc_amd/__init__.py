"""MI355X-native batched engine for the Data-Driven MPC per-timestep QP.

Public surface:
  * `DirectDataDrivenMPCController`, `DataDrivenMPCType`, `SlackVarConstraintTypes`
    -- drop-in mirror of the reference class
  * `BatchedDDMPC` -- batch of independent controller instances on one GPU
  * `utilities.hankel_matrix.{hankel_matrix, evaluate_persistent_excitation}`
The compute path is hand-written HIP behind the C ABI in include/ddmpc.h
(libddmpc.so, built by `python -m direct_data_driven_mpc_amd.build`).
"""
from .direct_data_driven_mpc_controller import (DataDrivenMPCType, DirectDataDrivenMPCController,
                                                 SlackVarConstraintTypes)
from .engine import BatchedDDMPC, hankel_matrix_batched

__all__ = ["DirectDataDrivenMPCController", "DataDrivenMPCType", "SlackVarConstraintTypes",
           "BatchedDDMPC", "hankel_matrix_batched"]
