import ctypes as C, numpy as np
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipHostRegister.argtypes=[C.c_void_p,C.c_size_t,C.c_uint]
hip.hipHostUnregister.argtypes=[C.c_void_p]
hip.hipGetErrorString.restype=C.c_char_p
hip.hipSetDevice(0)
a=np.zeros((1080,1920,3),np.uint8)
print("ptr %x"%a.ctypes.data, a.nbytes)
for flags in (0,1,2,3):
    r=hip.hipHostRegister(a.ctypes.data,a.nbytes,flags); print("flags",flags,r,hip.hipGetErrorString(r))
    if r==0: print(" unreg",hip.hipHostUnregister(a.ctypes.data))
p=a.ctypes.data & ~4095; n=(a.ctypes.data+a.nbytes+4095 & ~4095)-p
for flags in (0,1):
    r=hip.hipHostRegister(p,n,flags); print("aligned flags",flags,r,hip.hipGetErrorString(r))
    if r==0: print(" unreg",hip.hipHostUnregister(p))
b=np.zeros(300*301*3+5,np.uint8)[5:]
print("ptr %x"%b.ctypes.data)
for flags in (0,1):
    r=hip.hipHostRegister(b.ctypes.data,b.nbytes,flags); print("small flags",flags,r,hip.hipGetErrorString(r))
    if r==0: print(" unreg",hip.hipHostUnregister(b.ctypes.data))
