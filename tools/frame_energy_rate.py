"""Device time of k_frame_energy on the C2 batch (256 x 10 s, 16 kHz): HIP events via the engine's profiler."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth

clips = synth.synth_batch(256, 10.0, 16000, first=0)
eng = pkg.ProsodyEngine(0)
eng.upload(clips, 16000)
for requant in (False, True):
    for _ in range(5):
        eng.frame_energy_run(800, 800, requantize=requant)
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(50):
        eng.frame_energy_run(800, 800, requantize=requant)
    eng.sync()
    p = eng.profile()["k_frame_energy"]
    ms = p["total_ms"] / p["launches"]
    nbytes = sum(len(c) for c in clips) * 2 + 256 * 200 * 12
    print(f"requantize={requant}: {ms * 1e3:.1f} us per launch, {nbytes / ms / 1e6:.0f} GB/s of {nbytes / 1e6:.1f} MB algorithmic")
    eng.profile_enable(False)
eng.close()
