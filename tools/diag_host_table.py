"""Where does the time of materialising a multi-GB result table on the host go?  (config 5: [115, 5 M] fp64 = 4.6 GB)

Runs ON THE GPU BOX.  Measures, for one device block of --gb gigabytes:
  * np.empty destination + grx_download (what extract_features() does), three fresh destinations in a row
  * the same into an already faulted destination (the pipeline's own rate)
  * first touch of a fresh anonymous block: 1 thread, mmap + MADV_HUGEPAGE, MADV_POPULATE_WRITE split over threads
  * pinned destination: allocation time + direct copy
and prints the host's memory state (THP mode, cgroup limit, free memory) before and after.
"""
import argparse
import ctypes
import json
import mmap
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def host_state():
    out = {}
    for key, path in (('thp_enabled', '/sys/kernel/mm/transparent_hugepage/enabled'),
                      ('thp_defrag', '/sys/kernel/mm/transparent_hugepage/defrag'),
                      ('thp_shmem', '/sys/kernel/mm/transparent_hugepage/shmem_enabled'),
                      ('cgroup_memory_max', '/sys/fs/cgroup/memory.max'),
                      ('cgroup_memory_current', '/sys/fs/cgroup/memory.current'),
                      ('cgroup_memory_high', '/sys/fs/cgroup/memory.high')):
        try:
            out[key] = open(path).read().strip()
        except OSError as exc:
            out[key] = repr(exc)
    try:
        mi = {l.split(':')[0]: l.split(':')[1].strip() for l in open('/proc/meminfo')}
        out['meminfo'] = {k: mi[k] for k in ('MemTotal', 'MemFree', 'MemAvailable', 'Cached', 'Shmem', 'AnonHugePages') if k in mi}
    except OSError:
        pass
    out['cpus'] = os.cpu_count()
    try:
        out['affinity'] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return out


MADV_HUGEPAGE = 14
MADV_POPULATE_WRITE = 23
libc = ctypes.CDLL(None, use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]


def populate(addr, nbytes, threads):
    """MADV_POPULATE_WRITE of [addr, addr + nbytes) split over `threads` threads; returns seconds."""
    per = ((nbytes + threads - 1) // threads + (2 << 20) - 1) & ~((2 << 20) - 1)
    errs = []

    def run(i):
        off = i * per
        if off >= nbytes:
            return
        ln = min(per, nbytes - off)
        if libc.madvise(addr + off, ln, MADV_POPULATE_WRITE) != 0:
            errs.append(ctypes.get_errno())

    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(i,)) for i in range(threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return time.perf_counter() - t0, errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gb', type=float, default=4.6)
    args = ap.parse_args()
    from graphrole_amd import kernels as K
    nbytes = int(args.gb * 1e9) // 4096 * 4096
    res = {'bytes': nbytes, 'before': host_state()}
    dev = torch.empty(nbytes // 8, dtype=torch.float64, device='cuda')
    K._lib.call('grx_memset', K._ptr(dev), 1, nbytes, K._stream())
    torch.cuda.synchronize()

    # (1) what the product does: fresh np.empty + grx_download
    fresh = []
    keep = None
    for rep in range(3):
        t0 = time.perf_counter()
        out = K.to_host(dev)
        fresh.append(time.perf_counter() - t0)
        keep = out
    res['np_empty_download_s'] = fresh
    # (2) the pipeline's own rate: destination already faulted
    t0 = time.perf_counter()
    K._lib.call('grx_download', K._hptr(keep), K._ptr(dev), nbytes, K._stream())
    res['prefaulted_download_s'] = time.perf_counter() - t0
    del keep, out

    # (3) first touch alone
    t0 = time.perf_counter()
    a = np.empty(nbytes, np.uint8)
    a[::4096] = 1
    res['np_empty_touch_1thread_s'] = time.perf_counter() - t0
    del a
    for threads in (1, 8, 32):
        for huge in (False, True):
            mm = mmap.mmap(-1, nbytes + (2 << 20))
            buf = (ctypes.c_char * (nbytes + (2 << 20))).from_buffer(mm)
            addr = (ctypes.addressof(buf) + (2 << 20) - 1) & ~((2 << 20) - 1)
            if huge:
                libc.madvise(addr, nbytes, MADV_HUGEPAGE)
            dt, errs = populate(addr, nbytes, threads)
            res[f'populate_{threads}thr_{"huge" if huge else "base"}_s'] = dt
            if errs:
                res[f'populate_{threads}thr_{"huge" if huge else "base"}_errno'] = errs[:2]
            if threads == 32:
                # download into the populated block
                t0 = time.perf_counter()
                K._lib.call('grx_download', ctypes.c_void_p(addr), K._ptr(dev), nbytes, K._stream())
                res[f'download_into_populated_{"huge" if huge else "base"}_s'] = time.perf_counter() - t0
            del buf
            mm.close()

    # (4) pinned destination
    t0 = time.perf_counter()
    pin = torch.empty(nbytes // 8, dtype=torch.float64, pin_memory=True)
    res['pinned_alloc_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    pin.copy_(dev, non_blocking=True)
    torch.cuda.synchronize()
    res['pinned_copy_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    pin.copy_(dev, non_blocking=True)
    torch.cuda.synchronize()
    res['pinned_copy_again_s'] = time.perf_counter() - t0
    del pin
    # hipHostRegister of a populated pageable block
    mm = mmap.mmap(-1, nbytes + (2 << 20))
    buf = (ctypes.c_char * (nbytes + (2 << 20))).from_buffer(mm)
    addr = (ctypes.addressof(buf) + (2 << 20) - 1) & ~((2 << 20) - 1)
    libc.madvise(addr, nbytes, MADV_HUGEPAGE)
    dt, _ = populate(addr, nbytes, 32)
    hip = ctypes.CDLL('libamdhip64.so')
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(ctypes.c_void_p(addr), ctypes.c_size_t(nbytes), 0)
    res['host_register_s'] = time.perf_counter() - t0
    res['host_register_rc'] = rc
    if rc == 0:
        t0 = time.perf_counter()
        rc2 = hip.hipMemcpy(ctypes.c_void_p(addr), ctypes.c_void_p(dev.data_ptr()), ctypes.c_size_t(nbytes), 2)
        res['registered_copy_s'] = time.perf_counter() - t0
        res['registered_copy_rc'] = rc2
        t0 = time.perf_counter()
        hip.hipHostUnregister(ctypes.c_void_p(addr))
        res['host_unregister_s'] = time.perf_counter() - t0
    del buf
    mm.close()
    res['after'] = host_state()
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
