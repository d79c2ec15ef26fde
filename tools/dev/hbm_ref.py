"""Dev: what this box's HBM delivers to plain torch kernels (copy: read + write; sum: read only), for judging the STFT kernels' fractions."""
import torch
x = torch.empty(1 << 29, device="cuda", dtype=torch.float32).normal_()   # 2 GiB
y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
ms = t(lambda: y.copy_(x)); print(f"copy 2 GiB -> 2 GiB: {ms:.3f} ms, {2 * x.numel() * 4 / ms / 1e9:.2f} TB/s")
ms = t(lambda: x.sum()); print(f"sum 2 GiB: {ms:.3f} ms, {x.numel() * 4 / ms / 1e9:.2f} TB/s")
ms = t(lambda: y.fill_(1.0)); print(f"fill 2 GiB: {ms:.3f} ms, {x.numel() * 4 / ms / 1e9:.2f} TB/s")
