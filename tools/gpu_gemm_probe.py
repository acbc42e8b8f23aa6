"""Time the PPO MLP GEMM shapes (fp32) under both BLAS back ends and both weight layouts."""
import sys, time, torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10485
dev = "cuda:0"
shapes = [(39, 512), (512, 256), (256, 128), (128, 10), (168, 512), (128, 1)]
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    print("==== blas library:", lib)
    for (i, o) in shapes:
        x = torch.randn(B, i, device=dev); W = torch.randn(o, i, device=dev); b = torch.randn(o, device=dev)
        Wt = W.t().contiguous(); dy = torch.randn(B, o, device=dev)
        xt = x.t().contiguous(); dyt = dy.t().contiguous()
        gf = 2 * B * i * o / 1e9
        r = {}
        r["fwd F.linear(x,W)"] = bench(lambda: torch.nn.functional.linear(x, W, b))
        r["fwd addmm(b,x,Wt)"] = bench(lambda: torch.addmm(b, x, Wt))
        r["dX dy@W"] = bench(lambda: dy @ W)
        r["dX dy@Wt.T"] = bench(lambda: dy @ Wt.t())
        r["dW dy.T@x"] = bench(lambda: dy.t() @ x)
        r["dWt x.T@dy"] = bench(lambda: x.t() @ dy)
        r["dW dyt@x (dyt contiguous)"] = bench(lambda: dyt @ x)
        print(f"in {i:4d} out {o:4d} ({gf:5.2f} GF): " + "  ".join(f"{k}: {v:6.1f} us ({gf / v * 1e3:5.1f} TF)" for k, v in r.items()))
