"""Time the pitch analysis at another sample rate / floor (which selects the FFT path: see k_pitch_frames MODE)."""
import sys, time; sys.path.insert(0, '.')
import prosody_control_french_tts_amd as P
from prosody_control_french_tts_amd import synth
rate = int(sys.argv[1]) if len(sys.argv) > 1 else 44100
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
clips = [synth.synth_clip(i, 10.0, rate) for i in range(n)]
eng = P.ProsodyEngine(0); eng.upload(clips, rate)
sl = eng.whole_clip_slices(); params = P.PitchParams.praat(floor, 600.0)
eng.profile_enable(True)
for it in range(3):
    eng.profile_reset(); t = time.time(); eng.pitch_run(sl, params); eng.sync(); dt = time.time() - t
print("rate", rate, "floor", floor, "clips", n, "ms", dt * 1e3, {k: round(v["total_ms"], 3) for k, v in eng.profile().items()})
