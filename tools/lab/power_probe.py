"""Is the encoder power-limited?  The same launches (Whisper-small dims, 256 clips) with seeded random weights and with ALL-ZERO weights and
activations (zero LayerNorm gains: every GEMM / attention operand is zero, the instruction streams are identical, the datapaths do not toggle).
If the zero-data run is markedly faster the kernels run against the power limit, not against an issue or bandwidth limit.
usage: python tools/lab/power_probe.py   (prints per-kernel ms per encoder pass for both)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth, whisper_weights as WW

dims = WW.DIMS["small"]
clips = synth.synth_batch(256, 10.0, 16000)
eng = pkg.ProsodyEngine(0)
eng.upload(clips, 16000)
W = WW.synthetic_weights(dims)
for tag in ("random", "zeros", "random"):
    Wt = W if tag == "random" else {k: np.zeros_like(v) for k, v in W.items()}
    eng.whisper_load(dims, WW.pack(Wt, dims))
    eng.logmel_run(80); eng.whisper_encode_run(); eng.sync()
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(3):
        eng.logmel_run(80); eng.whisper_encode_run()
    eng.sync()
    pr = eng.profile(); eng.profile_enable(False)
    keep = {k: round(v["total_ms"] / 3, 2) for k, v in pr.items() if k in ("whisper_encoder", "k_gemm_flat:qkv", "k_gemm_flat:out", "k_gemm_flat:fc1", "k_gemm_flat:fc2",
                                                                            "k_attention_lean", "k_add_layernorm", "k_gemm_bf16")}
    print(f"{tag:>7}", keep, flush=True)
