"""wssdl_bus_amd.roi_pooling_layer -- MI355X counterpart of the reference's code/lib/roi_pooling_layer package (see wssdl_bus_amd/__init__.py)."""
