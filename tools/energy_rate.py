"""Device time of k_energy on the C2 batch (256 x 10 s, 16 kHz)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth
clips = synth.synth_batch(256, 10.0, 16000, first=0)
eng = pkg.ProsodyEngine(0)
eng.upload(clips, 16000); sl = eng.whole_clip_slices()
for _ in range(5): eng.energy_run(sl, 500)
eng.profile_enable(True); eng.profile_reset()
for _ in range(50): eng.energy_run(sl, 500)
eng.sync()
p = eng.profile()["k_energy"]; ms = p["total_ms"] / p["launches"]
print(f"k_energy {ms * 1e3:.1f} us per launch, {81.92 / ms / 1e3 * 1e3:.0f} GB/s")
eng.close()
