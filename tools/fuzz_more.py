"""More seeds of tests/test_gpu_fuzz.py than the suite runs (FUZZ_FROM / FUZZ_TO), both modes; prints the failing draws."""
import sys, os, traceback
sys.path.insert(0, os.getcwd())
import numpy as np
from pandora_amd.engine import Engine
from oracle import capi as orc
import tests.test_gpu_fuzz as fz
fails = 0
for lazy in (True, False):
    eng = Engine(0); eng.set_lazy(lazy); eng.lazy = lazy
    seeds = [int(x) for x in os.environ["FUZZ_SEEDS"].split(",")] if os.environ.get("FUZZ_SEEDS") else range(int(os.environ.get("FUZZ_FROM", "400")), int(os.environ.get("FUZZ_TO", "2400")))
    for seed in seeds:
        try:
            fz.test_random_pipeline_equals_oracle.__wrapped__(eng, orc, seed) if hasattr(fz.test_random_pipeline_equals_oracle, "__wrapped__") else fz.test_random_pipeline_equals_oracle(eng, orc, seed)
        except Exception as e:
            fails += 1
            c = fz.draw(seed)
            print("FAIL", "lazy" if lazy else "eager", seed, {k: v for k, v in c.items() if k not in ("L", "R", "masks", "grids")}, str(e)[:300].replace("\n", " "))
            if fails > 8:
                sys.exit(1)
    eng.close()
print("done, failures:", fails)
