"""wssdl_bus_amd.mil -- MI355X counterpart of the reference's code/lib/mil package (see wssdl_bus_amd/__init__.py)."""
