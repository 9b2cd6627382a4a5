"""Import-path compatibility with the reference package layout: callers do
`from direct_data_driven_mpc.direct_data_driven_mpc_controller import ...`
(utilities/controller/controller_creation.py:8-9 in the reference).  Everything
here re-exports `direct_data_driven_mpc_amd`."""
