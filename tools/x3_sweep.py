"""Sum of the ten layer sizes' forward / backward-data x3 kernel times under AUGCL_X3_ABL settings (tuning build)."""
import os, sys, torch, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, '.')
    from augcl_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libaugcl_tuning.so')
    L = _lib.load()
    dev = torch.device('cuda:0'); B = int(sys.argv[2])
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()
    w = torch.randn(32, 32, 3, 3, device=dev) * 0.05; bias = torch.zeros(32, device=dev)
    tot = [0.0, 0.0]
    for H in range(41, 21, -2):
        x = torch.rand(B, 32, H, H, device=dev); y = torch.empty(B, 32, H - 2, H - 2, device=dev)
        dz = torch.randn(B, 32, H - 2, H - 2, device=dev); dx = torch.empty(B, 32, H, H, device=dev)
        for k, fn in enumerate((lambda: _lib.call("augcl_enc_x3_conv32_forward", p(x), p(w), p(bias), p(y), B, H, H, 1, st),
                                lambda: _lib.call("augcl_enc_x3_conv32_backward_data", p(dz), p(w), p(x), p(dx), B, H - 2, H - 2, st))):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): fn()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            tot[k] += best
    print("fwd %.1f bwd %.1f" % tuple(tot))
else:
    B = sys.argv[1] if len(sys.argv) > 1 else "512"
    for abl in sys.argv[2:]:
        env = dict(os.environ, AUGCL_X3_ABL=abl)
        for rep in range(2):
            out = subprocess.run([sys.executable, __file__, "child", B], env=env, capture_output=True, text=True)
            print("ABL", abl, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
