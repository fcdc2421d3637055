"""Practical HBM write/copy ceilings on this GPU (torch fill / copy of 4 GiB), to price ksx_kernel's store rate."""
import torch
n = 4 << 30
a = torch.empty(n // 8, dtype=torch.float64, device="cuda")
b = torch.empty(n // 8, dtype=torch.float64, device="cuda")
for name, fn, bytes_moved in (("fill_", lambda: a.fill_(1.5), n), ("copy_", lambda: b.copy_(a), 2 * n),
                              ("zero_", lambda: a.zero_(), n)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%-6s 4 GiB: %.3f ms  -> %.2f TB/s (bytes moved %.1f GiB)" % (name, ms, bytes_moved / ms / 1e9, bytes_moved / 2**30))
