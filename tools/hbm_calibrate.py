"""What does this box's HBM deliver to simple streaming kernels?  (ceiling for roofline fractions)"""
import torch

dev = torch.device("cuda:0")
n = 3 * (1 << 28)  # 3 GiB of f32
a = torch.randn(n, device=dev)
b = torch.empty_like(a)


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in evs)
    return ms[len(ms) // 2]


gb = n * 4 / 1e9
t = timeit(lambda: b.copy_(a))
print("copy   %.2f GB r + %.2f GB w: %.3f ms -> %.0f GB/s total" % (gb, gb, t, 2 * gb / t * 1e3))
t = timeit(lambda: a.sum())
print("reduce %.2f GB r: %.3f ms -> %.0f GB/s" % (gb, t, gb / t * 1e3))
t = timeit(lambda: b.fill_(1.0))
print("fill   %.2f GB w: %.3f ms -> %.0f GB/s" % (gb, t, gb / t * 1e3))
t = timeit(lambda: torch.add(a, 1.0, out=b))
print("add    %.2f GB r + w: %.3f ms -> %.0f GB/s total" % (gb, t, 2 * gb / t * 1e3))
