"""Measured device copy bandwidth next to the nominal 8.0 TB/s: device-to-device copy of a 2 GiB buffer
(hipMemcpyAsync through torch), read + write bytes per second; and a read-only reduction for the read side alone."""
import json
import torch

assert torch.cuda.is_available()
n = 2 * 1024 ** 3 // 4
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
out = {}
for name, fn, nbytes in (("copy_d2d_read_plus_write", lambda: b.copy_(a), 2 * n * 4), ("read_only_sum", lambda: a.sum(), n * 4),
                         ("write_only_fill", lambda: b.fill_(1.0), n * 4)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out[name] = {"ms": ms, "GBps": nbytes / (ms * 1e-3) / 1e9}
out["device"] = torch.cuda.get_device_name(0)
print(json.dumps(out, indent=1))
