"""CPU oracle for the Pandora hot path - TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product package ``pandora_amd`` never does (tests/test_host_api.py::test_product_never_touches_the_oracle
enforces it).

``oracle.capi``  ctypes bindings to ``liboracle.so`` (plain-C restatement, oracle.c)
``oracle.ref``   loader for ``oracle/_ref/*.so`` - the reference's own pybind11 C++ modules compiled
                 from /root/reference by ``oracle/Makefile`` (present when built in the build
                 container; they travel to the GPU box as prebuilt binaries)
"""
