"""CPU oracle for the Pandora hot path - TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product package ``pandora_amd`` never does (tests/test_no_oracle_in_product.py
enforces it).

``oracle.capi``  ctypes bindings to ``liboracle.so`` (plain-C restatement, oracle.c)
``oracle.ref``   loader for ``oracle/_ref/*.so`` - the reference's own pybind11 C++ modules compiled
                 from /root/reference by ``oracle/Makefile`` (present when built in the build
                 container; they travel to the GPU box as prebuilt binaries)
"""
