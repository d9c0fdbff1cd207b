#!/usr/bin/env python3
"""where does a 12 GB file in /dev/shm go slowly?  write, pread (1 and 8 threads), mmap read, pinned H2D, D2H + write (GPU box)"""
import mmap, os, sys, threading, time
import numpy as np
import torch
GB = 1 << 30
n = int(float(sys.argv[1]) * GB) if len(sys.argv) > 1 else 12 * GB
fn = "/dev/shm/pfp_io_probe.bin"
buf = (np.arange(1 << 26, dtype=np.uint32) * np.uint32(2654435761) >> np.uint32(7)).astype(np.uint8)      # 64 MB of noise
t0 = time.perf_counter()
with open(fn, "wb") as fh:
    for off in range(0, n, len(buf)):
        fh.write(buf[: min(len(buf), n - off)].tobytes())
t = time.perf_counter() - t0
print(f"write {n / GB:.1f} GB (python, 64 MB pieces): {t:.2f} s = {n / t / 1e9:.2f} GB/s", flush=True)
fd = os.open(fn, os.O_RDONLY)
CH = 32 << 20
dst = np.empty(CH, dtype=np.uint8)
t0 = time.perf_counter()
for off in range(0, n, CH):
    os.preadv(fd, [memoryview(dst)[: min(CH, n - off)]], off)
t = time.perf_counter() - t0
print(f"pread, 1 thread, 32 MB pieces: {t:.2f} s = {n / t / 1e9:.2f} GB/s", flush=True)
def worker(k, T, out):
    d = np.empty(CH // T, dtype=np.uint8)
    for off in range(0, n, CH):
        ln = min(CH, n - off) // T
        os.preadv(fd, [memoryview(d)[:ln]], off + k * ln)
for T in (4, 8, 16):
    th = [threading.Thread(target=worker, args=(k, T, None)) for k in range(T)]
    t0 = time.perf_counter()
    [x.start() for x in th]; [x.join() for x in th]
    t = time.perf_counter() - t0
    print(f"pread, {T} threads: {t:.2f} s = {n / t / 1e9:.2f} GB/s", flush=True)
mm = mmap.mmap(fd, n, prot=mmap.PROT_READ)
t0 = time.perf_counter()
s = 0
for off in range(0, n, CH):
    s += int(np.frombuffer(mm, dtype=np.uint8, count=min(CH, n - off), offset=off)[::4096].sum())
t = time.perf_counter() - t0
print(f"mmap, one touch per page, 1 thread: {t:.2f} s = {n / t / 1e9:.2f} GB/s", flush=True)
pin = torch.empty(CH, dtype=torch.uint8).pin_memory()
dev = torch.empty(min(n, 16 * GB), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for off in range(0, dev.numel(), CH):
    dev[off:off + CH].copy_(pin[: min(CH, dev.numel() - off)], non_blocking=True)
torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"H2D from one pinned 32 MB buffer: {dev.numel() / t / 1e9:.2f} GB/s", flush=True)
t0 = time.perf_counter()
for off in range(0, dev.numel(), CH):
    pin[: min(CH, dev.numel() - off)].copy_(dev[off:off + CH], non_blocking=True)
torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"D2H into one pinned 32 MB buffer: {dev.numel() / t / 1e9:.2f} GB/s", flush=True)
os.close(fd)
os.unlink(fn)
print("cpus", len(os.sched_getaffinity(0)), "load", os.getloadavg())
