import ctypes, torch
torch.cuda.init()
lib = ctypes.CDLL("vfd_gan_amd/libvfdgan_hip.so")
arr = (ctypes.c_int * 16)()
n = lib.vfd_debug_occupancy(arr, 16)
print("occupancy:", list(arr)[:n])
