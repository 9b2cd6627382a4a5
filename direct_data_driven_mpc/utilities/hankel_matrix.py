"""Re-export of the GPU Hankel helpers under the reference's module path."""
from direct_data_driven_mpc_amd.utilities.hankel_matrix import (  # noqa: F401
    evaluate_persistent_excitation, hankel_matrix)
