import torch, time
torch.manual_seed(0)
dev = "cuda"
M = 192000
for (N, K, name) in [(2304, 768, "qkv"), (768, 768, "out"), (3072, 768, "fc1"), (768, 3072, "fc2")]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for form in ("nt",):
        f = (lambda: torch.matmul(a, w.t()))
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(name, form, M, N, K, f"{ms:.3f} ms", f"{2*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
