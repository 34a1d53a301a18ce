"""wssdl_bus_amd.nms -- MI355X counterpart of the reference's code/lib/nms package (see wssdl_bus_amd/__init__.py)."""
