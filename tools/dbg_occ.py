import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fldr-vfi_amd"))
import fldr_hip as hip
torch.zeros(1, device="cuda")
l = ctypes.CDLL(hip.LIB_PATH)
o = (ctypes.c_int * 4)()
print("rc", l.fldr_debug_conv_occupancy(o), list(o))
p = torch.cuda.get_device_properties(0)
print(p.name, "CUs", p.multi_processor_count, "shared/block", getattr(p, "shared_memory_per_block", None), "shared/mp", getattr(p, "shared_memory_per_multiprocessor", None))
