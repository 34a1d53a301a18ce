"""wssdl_bus_amd.networks -- MI355X counterpart of the reference's code/lib/networks package (see wssdl_bus_amd/__init__.py)."""
