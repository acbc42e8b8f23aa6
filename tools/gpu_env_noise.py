"""Is the box noisy?  Free-running loop of an unrelated ~75 us torch kernel, spin-drained: prints ms/iter per repeat."""
import time, torch
x = torch.zeros(48 * 1024 * 1024, device="cuda")   # 192 MB read+write per add_
for _ in range(50): x.add_(1.0)
torch.cuda.synchronize()
out = []
for rep in range(12):
    t0 = time.perf_counter()
    for _ in range(500): x.add_(1.0)
    ev = torch.cuda.Event(); ev.record()
    while not ev.query(): pass
    out.append((time.perf_counter() - t0) / 500 * 1e3)
print("ms/iter:", [round(o, 4) for o in out])
