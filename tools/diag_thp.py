import numpy as np, time, mmap, os
print(open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip(), '|', open('/sys/kernel/mm/transparent_hugepage/defrag').read().strip(), 'cpus', os.cpu_count())
n = 4_600_000_000
for rep in range(2):
    t0=time.perf_counter(); a=np.empty(n, np.uint8); a[::4096]=1; t1=time.perf_counter(); print('np.empty touch 1 thread', round(t1-t0,3)); del a
mm = mmap.mmap(-1, n); 
try:
    mm.madvise(mmap.MADV_HUGEPAGE)
except Exception as e: print('madvise', e)
b=np.frombuffer(mm, np.uint8)
t0=time.perf_counter(); b[::4096]=1; t1=time.perf_counter(); print('mmap+MADV_HUGEPAGE touch', round(t1-t0,3))
print([l for l in open('/proc/meminfo') if 'AnonHuge' in l or 'MemFree' in l])
