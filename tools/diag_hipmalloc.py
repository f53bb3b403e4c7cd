"""How long does a first device allocation of several GB take (the arena of grx_refex_run at config 5 is ~7 GB)?
Raw hipMalloc / hipFree, torch's caching allocator cold and warm, and the first touch of the block."""
import ctypes, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
hip = ctypes.CDLL('libamdhip64.so')
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
res = {}
for gb in (1, 4, 8, 16):
    p = ctypes.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gb << 30)); t1 = time.perf_counter()
    rc2 = hip.hipMemset(p, 0, ctypes.c_size_t(gb << 30)); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    rc3 = hip.hipMemset(p, 0, ctypes.c_size_t(gb << 30)); hip.hipDeviceSynchronize(); t3 = time.perf_counter()
    hip.hipFree(p); t4 = time.perf_counter()
    res[f'raw_{gb}GiB'] = {'malloc_s': t1 - t0, 'first_memset_s': t2 - t1, 'second_memset_s': t3 - t2, 'free_s': t4 - t3, 'rc': [rc, rc2, rc3]}
for rep in range(3):
    t0 = time.perf_counter(); a = torch.empty(7_500_000_000, dtype=torch.uint8, device='cuda'); torch.cuda.synchronize(); t1 = time.perf_counter()
    b = torch.empty(7_500_000_000, dtype=torch.uint8, device='cuda'); torch.cuda.synchronize(); t2 = time.perf_counter()
    del a, b
    res[f'torch_rep{rep}'] = {'first_7.5GB_s': t1 - t0, 'second_7.5GB_while_first_alive_s': t2 - t1}
print(json.dumps(res, indent=1))
