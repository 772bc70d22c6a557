"""GPU box: what a plain streaming kernel gets out of HBM on this part (torch elementwise kernels, 16-bit elements):
read-only (sum), write-only (fill), copy (1 read + 1 write), and the 1 : 4 read : write mix of a 256 -> 1024 1x1 layer --
at sizes that do not fit the 256 MB Infinity Cache, and at the sizes of one layer's maps (which do, when the same buffers are
reused back to back)."""
import torch

def timed(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for mb in (26, 105, 1024, 4096):
    n = mb * 2 ** 20 // 2
    x = torch.randn(n, device="cuda").to(torch.bfloat16) if mb <= 1024 else torch.zeros(n, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(x)
    t = timed(lambda: y.fill_(1.0))
    print("%5d MB  write-only (fill)  %.2f TB/s" % (mb, mb * 2 ** 20 / t / 1e12))
    t = timed(lambda: y.copy_(x))
    print("%5d MB  copy               %.2f TB/s (read + write bytes)" % (mb, 2 * mb * 2 ** 20 / t / 1e12))
    t = timed(lambda: torch.sum(x, dtype=torch.float32))
    print("%5d MB  read-only (sum)    %.2f TB/s" % (mb, mb * 2 ** 20 / t / 1e12), flush=True)
# rotating buffers (nothing is re-used before 2 GB of other traffic): the 26 MB in / 105 MB out pattern of the expand layer
xs = [torch.zeros(26 * 2 ** 19, device="cuda", dtype=torch.bfloat16) for _ in range(16)]
ys = [torch.empty(105 * 2 ** 19, device="cuda", dtype=torch.bfloat16) for _ in range(16)]
i = [0]
def mix():
    k = i[0] % 16
    i[0] += 1
    ys[k].view(4, -1)[:] = xs[k]          # broadcast copy: reads 26 MB, writes 105 MB
t = timed(mix, 32)
print("rotating 26 MB in -> 105 MB out (broadcast copy): %.1f us, %.2f TB/s" % (t * 1e6, 131 * 2 ** 20 / t / 1e12))
