"""Microbenchmark formulations of the projection GEMM Y = X W^T (X: B x 14112, W: 100 x 14112), forward + backward."""
import torch, torch.nn.functional as F
dev = torch.device('cuda:0')
K, N = 14112, 100
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
for B in (512, 256):
    x = torch.randn(B, K, device=dev, requires_grad=True); w = (torch.randn(N, K, device=dev) * 0.01).requires_grad_(True)
    b = torch.zeros(N, device=dev, requires_grad=True); g = torch.randn(B, N, device=dev)
    forms = {
        "F.linear": lambda: F.linear(x, w, b),
        "(W X^T)^T": lambda: torch.mm(w, x.t()).t() + b,
        "splitK16 bmm": lambda: torch.bmm(x.view(B, 16, K // 16).transpose(0, 1), w.view(N, 16, K // 16).permute(1, 2, 0)).sum(0) + b,
        "splitK32 bmm": lambda: torch.bmm(x.view(B, 32, K // 32).transpose(0, 1), w.view(N, 32, K // 32).permute(1, 2, 0)).sum(0) + b,
        "splitK8 bmm": lambda: torch.bmm(x.view(B, 8, K // 8).transpose(0, 1), w.view(N, 8, K // 8).permute(1, 2, 0)).sum(0) + b,
    }
    ref = forms["F.linear"]().detach()
    for name, f in forms.items():
        y = f(); err = float((y.detach() - ref).abs().max() / ref.abs().max())
        with torch.no_grad(): tf = timeit(f)
        def fb():
            x.grad = w.grad = b.grad = None
            f().backward(g)
        tfb = timeit(fb)
        print("B=%d %-14s fwd %6.1f us   fwd+bwd %6.1f us   rel err %.1e" % (B, name, tf, tfb, err))
