"""wssdl_bus_amd.rpn_msr -- MI355X counterpart of the reference's code/lib/rpn_msr package (see wssdl_bus_amd/__init__.py)."""
