import sys, time, ctypes as C; sys.path.insert(0,'/root/repo')
import numpy as np
from v2xgnn.rl import native_sim as ns
lib=ns._load()
key=np.random.get_state()[1].copy(); p=np.array([624],np.int32)
for n in (200000, 1000000):
    t0=time.perf_counter()
    for _ in range(5): lib.v2xsim_np_shuffle_skip(key.ctypes.data_as(C.POINTER(C.c_uint32)), p.ctypes.data_as(C.POINTER(C.c_int32)), n)
    t1=time.perf_counter()
    for _ in range(5): ns.np_choice_noreplace(n, 4096)
    t2=time.perf_counter()
    for _ in range(3): np.random.choice(n, 4096, replace=False)
    t3=time.perf_counter()
    print(n, "draws only (branchy) %.2f ms | native choice %.2f ms | numpy %.2f ms" % ((t1-t0)/5*1e3, (t2-t1)/5*1e3, (t3-t2)/3*1e3))
