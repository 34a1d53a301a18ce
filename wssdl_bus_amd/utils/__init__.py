"""wssdl_bus_amd.utils -- MI355X counterpart of the reference's code/lib/utils package (see wssdl_bus_amd/__init__.py)."""
