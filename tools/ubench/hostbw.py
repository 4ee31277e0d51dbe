import numpy as np, time, ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
from pandora_amd import _lib
L=_lib.lib()
g=np.full((2048,2048),5,np.int64)
mn=C.c_int64(); mx=C.c_int64()
ts=[]
for _ in range(20):
    t=time.perf_counter(); L.pmx_host_minmax_i64(g.ctypes.data_as(_lib.c_i64_p), g.size, C.byref(mn), C.byref(mx)); ts.append(time.perf_counter()-t)
a=np.random.rand(2048,2048).astype(np.float32)
tf=[]
for _ in range(20):
    t=time.perf_counter(); L.pmx_host_fingerprint(a.ctypes.data, a.nbytes); tf.append(time.perf_counter()-t)
print(os.environ.get("PMX_HOST_THREADS"), "minmax ms", min(ts)*1e3, "fingerprint ms", min(tf)*1e3, os.cpu_count())
