"""Control for race_probe*.py: plain PyTorch kernels (rocBLAS GEMM, elementwise, reductions, softmax) repeated by several processes at once on one
GPU, outputs hashed against iteration 0.  usage: PROBE_PAR=3 race_probe_torch.py iters"""
import hashlib, os, subprocess, sys
if os.environ.get("PROBE_PAR") and len(sys.argv) < 3:
    n = int(os.environ["PROBE_PAR"]); env = {k: v for k, v in os.environ.items() if k != "PROBE_PAR"}
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), sys.argv[1] if len(sys.argv) > 1 else "40", f"p{k}"], env=env) for k in range(n)]
    sys.exit(max(p.wait() for p in ps))
import torch
iters = int(sys.argv[1]); tag = sys.argv[2] if len(sys.argv) > 2 else "solo"
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn(4096, 1024, device="cuda", generator=g, dtype=torch.float16)
b = torch.randn(1024, 2048, device="cuda", generator=g, dtype=torch.float16)
x = torch.randn(3, 80, 3000, device="cuda", generator=g)
h = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:10]
ref, bad = None, {}
for it in range(iters):
    out = {}
    c = a @ b; out["gemm"] = h(c)
    out["softmax"] = h(torch.softmax(c.float(), dim=-1))
    out["ln"] = h(torch.nn.functional.layer_norm(c.float(), (2048,)))
    out["conv"] = h(torch.nn.functional.conv1d(x, torch.ones(16, 80, 3, device="cuda") / 240, padding=1))
    out["stft"] = h(torch.view_as_real(torch.stft(x[0, 0], 400, 160, window=torch.hann_window(400, device="cuda"), return_complex=True)))
    if ref is None: ref = out
    for k in out:
        if out[k] != ref[k]: bad[k] = bad.get(k, 0) + 1
print(tag, "torch iterations", iters, "ops that ever differed from iteration 0:", bad or "none")
