import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1] if len(sys.argv) > 1 else "torch_first"
def maps():
    s = set()
    for l in open("/proc/self/maps"):
        if any(k in l for k in ("amdhip64", "hsa-runtime", "pysparse_hip")):
            s.add(l.split()[-1])
    return sorted(s)
if order == "torch_first":
    import torch
    print("torch cuda", torch.cuda.is_available(), torch.cuda.device_count()); torch.zeros(4, device="cuda")
    from pysparse_amd import _capi
    L = _capi.lib()
    print("psp count", L.psp_device_count(), "set", L.psp_set_device(0), L.psp_last_error())
else:
    from pysparse_amd import _capi
    L = _capi.lib()
    print("psp count", L.psp_device_count(), "set", L.psp_set_device(0), L.psp_last_error())
    import torch
    print("torch cuda", torch.cuda.is_available(), torch.cuda.device_count()); print(torch.zeros(4, device="cuda"))
print("\n".join(maps()))
