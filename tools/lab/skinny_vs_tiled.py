"""Bitwise comparison of the few-row GEMM kernel (k_gemm_skinny) with the 128 x 128 kernel on the same operands, per operand type and epilogue.
Run twice (PCE_GEMM_SKINNY=1 / 0) and compare the saved arrays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import prosody_control_french_tts_amd as pkg
out = sys.argv[1]
eng = pkg.ProsodyEngine(0)
rng = np.random.default_rng(0)
res = {}
for ops in ("bf16", "fp16"):
    eng.whisper_set_operands(ops)
    for (M, N, K) in ((256, 768, 768), (256, 3072, 768), (256, 768, 3072), (200, 768, 768)):
        A = rng.standard_normal((M, K)).astype(np.float32); B = (rng.standard_normal((N, K)) * 0.05).astype(np.float32); bias = rng.standard_normal(N).astype(np.float32)
        for epi in (16, 17, 19):
            res[f"{ops}_{M}_{N}_{K}_{epi}"] = eng.selftest_gemm(A, B, bias, epi)
np.savez(out, **res)
eng.close()
