"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, bf16 in, f32 accumulate) reaches on the GEMM shapes of the
conv layers at batch 64 (M = pixels, N = Cout, K = k*k*Cin): a practical MFMA ceiling to hold the conv kernels against.
The conv kernels do MORE than this GEMM (im2col / halo gather, bias, activation, residual), so this is an upper reference."""
import torch
shapes = [("3x3 128->256 @52", 173056, 256, 1152), ("3x3 256->512 @26", 43264, 512, 2304), ("3x3 512->1024 @13", 10816, 1024, 4608),
          ("3x3 64->128 @104", 692224, 128, 576), ("1x1 256->128 @52", 173056, 128, 256), ("1x1 512->256 @26", 43264, 256, 512),
          ("1x1 1024->512 @13", 10816, 512, 1024), ("big square", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    R = max(2, min(16, int(3e9 // ((M * K + K * N + M * N) * 2))))      # ring of operand sets larger than L2 + MALL
    a = [torch.randn(M, K, device="cuda", dtype=torch.bfloat16) for _ in range(R)]
    b = [torch.randn(K, N, device="cuda", dtype=torch.bfloat16) for _ in range(R)]
    c = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(R)]
    for i in range(3):
        torch.matmul(a[i % R], b[i % R], out=c[i % R])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for i in range(n):
        torch.matmul(a[i % R], b[i % R], out=c[i % R])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%-20s M=%6d N=%4d K=%4d  %8.1f us  %7.1f TFLOP/s" % (name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
    del a, b, c
