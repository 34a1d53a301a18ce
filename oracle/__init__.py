"""CPU oracle for the wssdl_bus detection hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/ (incl. the checking scripts under tools/ that tests/test_gpu_fuzz.py and ``tools/roofline_leg.py --check``
run: they compare, they are never measured or shipped), ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package, and only as the checker.  The product package
``wssdl_bus_amd`` never imports it.

  np_oracle.py        NumPy restatement of the reference's Python layers
  c/oracle_kernels.c  C restatement of the reference's native kernels
  c_oracle.py         ctypes binding of the above
  build_ref.py        recipe that compiles the reference's own Cython kernels
                      from /root/reference into oracle/_ref/ (git-ignored)
  ref_kernels.py      loader for oracle/_ref/*.so
  ref_python_stage.py stages + imports the reference's Python layers (build
                      container only; used to generate tests/golden/*.npz)
"""
