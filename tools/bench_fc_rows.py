"""fc_rows (fc6 / fc7 / head 1x1 products) at the bench shapes: train-mode capacity 3024 with 684 (the bench since round 4), 468 and 1500 live rows, test-mode
capacity 336 with 75 live rows (split-K), batch-1 capacity 21, and the tall head shapes. Prints ms and TFLOP/s."""
import sys, torch
sys.path.insert(0, '/root/repo')
from posecnn_amd import ops
dev = torch.device('cuda:0')
for (M, K, N, cnt) in ((3024, 25088, 4096, 684), (3024, 4096, 4096, 684), (3024, 25088, 4096, 468), (3024, 25088, 4096, 1500), (336, 25088, 4096, 75), (21, 25088, 4096, 5), (76800, 512, 64, None), (76800, 512, 128, None)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.01; b = torch.randn(N, device=dev)
    c = None if cnt is None else torch.tensor([cnt], dtype=torch.int32, device=dev)
    for _ in range(3): y = ops.fc_rows(x, w, b, True, num_rows=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y = ops.fc_rows(x, w, b, True, num_rows=c)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    rows = M if cnt is None else cnt
    print(M, K, N, cnt, "%.3f ms  %.1f TFLOP/s" % (ms, 2.0 * rows * K * N / ms / 1e9))
    del x, w
