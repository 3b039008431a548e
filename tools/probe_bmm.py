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
for name, T, K, N in [("conv4_1", 19200, 256, 512), ("conv4_2", 19200, 512, 512), ("conv5", 4800, 512, 512), ("conv3_2", 76800, 256, 256)]:
    V = torch.randn((16, T, K), device=dev)
    U = torch.randn((16, K, N), device=dev)
    M = torch.empty((16, T, N), device=dev)
    ms = timeit(lambda: torch.bmm(V, U, out=M))
    fl = 2.0 * 16 * T * K * N
    # one big GEMM alternative: [T, 16K] block structure not applicable; also try strided-batched via matmul
    res[name] = {"bmm_ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1)}
    # same flops as a single GEMM (upper bound of the library's fp32 rate at this size)
    A = torch.randn((16 * T, K), device=dev); Bm = torch.randn((K, N), device=dev)
    ms1 = timeit(lambda: torch.mm(A, Bm))
    res[name]["single_gemm_ms"] = round(ms1, 4); res[name]["single_TFLOPs"] = round(fl / ms1 / 1e9, 1)
print(json.dumps(res))
