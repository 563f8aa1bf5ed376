"""What the vendor GEMM (hipBLASLt through torch.matmul) sustains on this GPU for bf16 with float32 accumulation, on random and
on all-zero operands: the practical ceiling that the encoder's implicit-GEMM kernels are held against.  Not part of the product."""
import torch


def rate(M, K, N, zero, reps=30):
    a = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16) if zero else torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.zeros(K, N, device="cuda", dtype=torch.bfloat16) if zero else torch.randn(K, N, device="cuda").to(torch.bfloat16)
    for _ in range(5):
        a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        a @ b
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return 2.0 * M * K * N / ms / 1e9, ms


for shape in ((100352, 4608, 512), (401408, 2304, 256), (8192, 8192, 8192), (6422528, 576, 64)):
    for zero in (False, True):
        t, ms = rate(*shape, zero)
        print("M=%d K=%d N=%d %s: %.0f TFLOP/s (%.3f ms)" % (*shape, "zeros " if zero else "random", t, ms))
