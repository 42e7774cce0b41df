import time, sys, os
t0=time.perf_counter()
sys.path.insert(0, os.getcwd())
import ctypes
t1=time.perf_counter()
import numpy
t2=time.perf_counter()
from impop_amd import _lib
lib=_lib.load()
t3=time.perf_counter()
h=ctypes.c_void_p()
rc=lib.impop_ctx_create(0,None,ctypes.byref(h))
t4=time.perf_counter()
print("ctypes %.3f numpy %.3f libload %.3f ctx_create %.3f rc=%d"%(t1-t0,t2-t1,t3-t2,t4-t3,rc))
