"""Re-export of the MI355X controller under the reference's module path."""
from direct_data_driven_mpc_amd.direct_data_driven_mpc_controller import (  # noqa: F401
    DataDrivenMPCType, DirectDataDrivenMPCController, SlackVarConstraintTypes)
