"""The heads' GEMM shapes: hand-written augcl_gemm_f32 vs the GEMM library through torch (us per launch, hipGraph replay of 20
back-to-back launches so that host time is out of the picture).  python tools/gemm_bench.py"""
import sys
import torch
sys.path.insert(0, '.')
from augcl_amd import head_ops  # noqa: E402

dev = torch.device('cuda:0')
SHAPES = [  # name, M, N, K, batch, a_rows, b_rows
    ("twin L1 fwd  (512x106 -> 2048)", 512, 2048, 106, 1, 0, 0), ("twin L2 fwd  2x(512x1024x1024)", 512, 1024, 1024, 2, 0, 0),
    ("twin L2 fwd  2x(256x1024x1024)", 256, 1024, 1024, 2, 0, 0), ("twin dz1     2x(512x1024x1024) NN", 512, 1024, 1024, 2, 0, 1),
    ("twin dx      512x2048 -> 106", 512, 106, 2048, 1, 0, 1), ("twin dW1     2048x512 . 512x106", 2048, 106, 512, 1, 1, 1),
    ("twin dW2     2x(1024x512 . 512x1024)", 1024, 1024, 512, 2, 1, 1), ("actor L1 fwd 256x100 -> 1024", 256, 1024, 100, 1, 0, 0),
    ("actor L2 fwd 256x1024x1024", 256, 1024, 1024, 1, 0, 0), ("actor L3 fwd 256x1024 -> 12", 256, 12, 1024, 1, 0, 0),
    ("actor dz2    256x12 -> 1024", 256, 1024, 12, 1, 0, 1), ("actor dW3    12x256 . 256x1024", 12, 1024, 256, 1, 1, 1),
    ("proj dW      100x512 . 512x14112", 100, 14112, 512, 1, 1, 1), ("proj dW      100x256 . 256x14112", 100, 14112, 256, 1, 1, 1),
]


def graph_time(fn, n=20, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


tot = [0.0, 0.0]
for name, M, N, K, batch, ar, br in SHAPES:
    bs = (batch,) if batch > 1 else ()
    a = torch.randn(*bs, *((K, M) if ar else (M, K)), device=dev)
    b = torch.randn(*bs, *((K, N) if br else (N, K)), device=dev)
    A32 = a.transpose(-1, -2) if ar else a
    B32 = b if br else b.transpose(-1, -2)
    t_lib = graph_time(lambda: torch.matmul(A32, B32))
    t_hip = graph_time(lambda: head_ops.gemm(a, b, a_rows=bool(ar), b_rows=bool(br)))
    tot[0] += t_lib; tot[1] += t_hip
    print(f"{name:40s} lib {t_lib:7.2f} us   hip {t_hip:7.2f} us   ({2 * M * N * K * batch / t_hip / 1e6:7.1f} TFLOP/s f32-equivalent)", flush=True)
print(f"{'sum':40s} lib {tot[0]:7.2f} us   hip {tot[1]:7.2f} us")
