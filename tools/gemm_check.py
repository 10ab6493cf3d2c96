"""Debug aid: the persistent 256 x 256 GEMM against torch on a few shapes / epilogues."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prosody_control_french_tts_amd as pkg
rng = np.random.default_rng(0)
with pkg.ProsodyEngine(0) as eng:
    for (M, N, K, epi) in [(3000, 1536, 768, 0), (3000, 768, 768, 0), (3000, 3072, 768, 1), (6000, 768, 3072, 0), (3000, 768, 768, 2)]:
        A = rng.standard_normal((M, K)).astype(np.float32); B = (rng.standard_normal((N, K)) * 0.05).astype(np.float32); bias = rng.standard_normal(N).astype(np.float32)
        a = torch.from_numpy(A).to(torch.bfloat16).float(); b = torch.from_numpy(B).to(torch.bfloat16).float()
        ref = (a @ b.T + torch.from_numpy(bias)).numpy()
        if epi == 1: ref = torch.nn.functional.gelu(torch.from_numpy(ref)).numpy()
        got = eng.selftest_gemm(A, B, bias, epi, 1500, 1536)
        if epi == 2:
            g2 = np.zeros_like(ref)
            for c in range(M // 1500): g2[c * 1500:(c + 1) * 1500] = got[c, :, :1500].T
            pad = got[:, :, 1500:]
            print("  pad columns untouched:", float(np.abs(pad).max()))
            got = g2
        bad = ~np.isfinite(got)
        err = np.abs(np.where(bad, 0, got) - ref)
        print((M, N, K, epi), "nan", int(bad.sum()), "max err", float(err.max()), "rel", float(np.linalg.norm(err) / np.linalg.norm(ref)), "worst rows", np.unique(np.where(err > 0.1)[0])[:8], "cols", np.unique(np.where(err > 0.1)[1])[:8])
    # structure of the wrong elements of a plain case
    M, N, K = 3000, 768, 768
    A = rng.standard_normal((M, K)).astype(np.float32); B = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    a = torch.from_numpy(A).to(torch.bfloat16).float(); b = torch.from_numpy(B).to(torch.bfloat16).float()
    ref = (a @ b.T).numpy()
    got = eng.selftest_gemm(A, B, None, 0)
    bad = ~(np.abs(got - ref) < 0.1)
    r, c = np.where(bad)
    print("bad count", bad.sum(), "of", bad.size)
    print("rows mod 256:", np.unique(r % 256)); print("cols mod 256:", np.unique(c % 256)); print("row tiles:", np.unique(r // 256), "col tiles:", np.unique(c // 256))
    import collections
    print("values sample:", got[bad][:8])
