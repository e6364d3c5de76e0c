# what the memory system gives: 16 GB of pure writes, a 16 GB -> 16 GB copy, a strided read of 96 of every 288 bytes
import torch, time
dev = torch.device("cuda", 0)
n = 4_000_000_000
a = torch.empty(n, dtype=torch.int32, device=dev)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
dt = t(lambda: a.fill_(1)); print(f"fill 16 GB: {dt*1e3:.2f} ms  {16/dt/1e3:.2f} TB/s written")
b = torch.empty_like(a)
dt = t(lambda: b.copy_(a)); print(f"copy 16 GB: {dt*1e3:.2f} ms  {32/dt/1e3:.2f} TB/s moved")
del b
# gather 96 B of every 288 B record (24 of 72 int32) -> like K4's SEQ fetch; output small
v = a[: (n // 72) * 72].view(-1, 72)
dt = t(lambda: v[:, 12:36].sum(dim=1)); print(f"read 96 of every 288 B over 16 GB: {dt*1e3:.2f} ms  useful {n*4/3/dt/1e12:.2f} TB/s, lines {16/dt/1e3:.2f} TB/s if whole lines are fetched")
