"""wssdl_bus_amd.fast_rcnn -- MI355X counterpart of the reference's code/lib/fast_rcnn package (see wssdl_bus_amd/__init__.py)."""
