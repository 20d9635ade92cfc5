"""Device bandwidth probe (not a test): copy and read-only streams through slam_copy_probe_dev."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "slam-process_amd"))
import torch
from slamhip import _lib
c = _lib.ctx(0)
n = 4 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
b = torch.empty(n, dtype=torch.uint8, device="cuda")
c.set_stream(torch.cuda.current_stream().cuda_stream)
for name, dst, src, traffic in (("copy", b, a, 2 * n), ("read", a, a, n)):
    for _ in range(3):
        c.check(c.lib.slam_copy_probe_dev(c.h, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        c.check(c.lib.slam_copy_probe_dev(c.h, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {traffic / dt / 1e12:.2f} TB/s ({dt * 1e3:.3f} ms per {n >> 20} MiB)")

lib = c.lib
lib.slam_read_probe_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
for mode in (1, 2, 3):
    for _ in range(3):
        c.check(lib.slam_read_probe_dev(c.h, C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), n, mode))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        c.check(lib.slam_read_probe_dev(c.h, C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), n, mode))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"tile-shaped read mode {mode}: {n / dt / 1e12:.2f} TB/s")

for mode, traffic in ((10, 2 * n), (11, 2 * n), (12, 2 * n), (20, n + n // 3), (21, n + n // 3), (22, n + n // 3)):
    for _ in range(3):
        c.check(lib.slam_read_probe_dev(c.h, C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), n, mode))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        c.check(lib.slam_read_probe_dev(c.h, C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), n, mode))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"read:write {'1:1' if mode < 20 else '3:1'} mode {mode % 10} (0 plain, 1 nt stores, 2 nt loads+stores): "
          f"{traffic / dt / 1e12:.2f} TB/s total")
