import torch, json
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]
res = {}
import os
if os.environ.get("PROBE_BLAS"):
    torch.backends.cuda.preferred_blas_library(os.environ["PROBE_BLAS"])
print("blas:", torch.backends.cuda.preferred_blas_library())
NB = int(os.environ.get("PROBE_PLANES", "36"))   # 16: F(2x2,3x3), 36: F(4x4,3x3) with T/4 tiles
for name, T, K, N in [("conv2_2", 19200 * 4, 128, 128), ("conv3_2", 19200, 256, 256), ("conv4_1", 4800, 256, 512), ("conv4_2", 4800, 512, 512), ("conv5", 1280, 512, 512)]:
    V = torch.randn((NB, T, K), device=dev)
    U = torch.randn((NB, K, N), device=dev)
    M = torch.empty((NB, T, N), device=dev)
    ms = timeit(lambda: torch.bmm(V, U, out=M))
    fl = 2.0 * NB * T * K * N
    # one big GEMM alternative: [T, 16K] block structure not applicable; also try strided-batched via matmul
    res[name] = {"bmm_ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1)}
    # same flops as a single GEMM (upper bound of the library's fp32 rate at this size)
    A = torch.randn((NB * T, K), device=dev); Bm = torch.randn((K, N), device=dev)
    ms1 = timeit(lambda: torch.mm(A, Bm))
    res[name]["single_gemm_ms"] = round(ms1, 4); res[name]["single_TFLOPs"] = round(fl / ms1 / 1e9, 1)
print(json.dumps(res))
