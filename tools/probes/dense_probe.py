"""Timing of the dense float64 primitives a Gram-based PCA solver would use (not a test)."""
import time
import torch

def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

dev = "cuda"
for g in (2000, 4000):
    a = torch.randn(g, g, dtype=torch.float64, device=dev); c = a @ a.T / g
    z = torch.randn(g, 128, dtype=torch.float64, device=dev)
    z64 = z[:, :64].contiguous()
    print(f"g={g}")
    print("  eigh(g x g) f64        %.2f ms" % t(lambda: torch.linalg.eigh(c), 2))
    print("  C @ Z (g x g x 128)    %.3f ms" % t(lambda: c @ z, 20))
    print("  C @ Z (g x g x 64)     %.3f ms" % t(lambda: c @ z64, 20))
    print("  C @ C                  %.3f ms" % t(lambda: c @ c, 5))
    print("  Z^T Z (128)            %.3f ms" % t(lambda: z.T @ z, 20))
    s = z.T @ z
    print("  cholesky(128)          %.3f ms" % t(lambda: torch.linalg.cholesky(s), 20))
    l = torch.linalg.cholesky(s)
    print("  trsm (g x 128)         %.3f ms" % t(lambda: torch.linalg.solve_triangular(l, z.T, upper=False), 20))
    print("  qr(g x 128)            %.3f ms" % t(lambda: torch.linalg.qr(z), 5))
    print("  eigh(128)              %.3f ms" % t(lambda: torch.linalg.eigh(s), 10))
    s5 = torch.randn(512, 512, dtype=torch.float64, device=dev); s5 = s5 @ s5.T
    print("  eigh(512)              %.3f ms" % t(lambda: torch.linalg.eigh(s5), 5))
    cf = c.float(); zf = z.float()
    print("  f32 C @ Z (128)        %.3f ms" % t(lambda: cf @ zf, 20))
    sc = s.cpu()
    t0 = time.perf_counter(); torch.linalg.eigh(sc); print("  cpu eigh(128)          %.3f ms" % ((time.perf_counter() - t0) * 1e3))
