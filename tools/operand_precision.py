"""bf16 vs fp16 operands vs fp16 operands + fp16 residual stream at full depth (Whisper-small, 12 + 12 layers, fixed-seed random-init weights: no trained checkpoint exists
offline) against the float32 restatement (oracle/whisper_oracle.py):
  * encoder output: relative L2 and maximum error, 4 ten-second clips
  * alignment cost matrix: relative L2; word-boundary frames identical / within one 20 ms step of the fp32 path
  * greedy decisions: teacher-forced along the fp32 restatement's own greedy sequences, EVERY step counted (no margin filter):
    fraction of steps where the engine's choice differs from the restatement's, with the fp32 top-2 margin of those steps
usage: operand_precision.py [n_decision_clips] [steps]      (writes one JSON object to stdout)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth, whisper_weights as WW
from prosody_control_french_tts_amd.Aligners import decoding as DEC
from oracle import whisper_oracle as WO

n_dec = int(sys.argv[1]) if len(sys.argv) > 1 else 48
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 44
dims, tdims = WW.DIMS["small"], dict(WW.TEXT_DIMS["small"], n_vocab=2048)
W = WW.synthetic_weights(dims)
Wa = WW.synthetic_decoder_weights(tdims)            # alignment leg (as tests/test_gpu_whisper.py)
Wg = WW.greedy_test_decoder_weights(tdims, seed=79) # decision leg: a decoder whose next token depends on position and audio
rules = dict(eot=1990, no_timestamps=2003, timestamp_begin=2004, suppress_tokens=list(range(1991, 2003)), blank_tokens=[220, 1990], max_initial_timestamp_index=20)
init = [1991, 1995, 2000]
clips4 = [synth.synth_clip(40 + i, seconds=10.0) for i in range(4)]
clipsd = [synth.synth_clip(100 + i, seconds=6.0) for i in range(n_dec)]
rng = np.random.default_rng(33)
toks = [rng.integers(0, 1990, size=int(n)).tolist() for n in (24, 31, 40, 47)]
frames = [len(c) // 160 for c in clips4]
t0 = time.time()
ref_enc = [WO.encoder_forward(WO.log_mel(c, 80), W, dims) for c in clips4]
ref_al = [WO.find_alignment(toks[i], ref_enc[i], Wa, tdims, frames[i], 3) for i in range(4)]
ref_encd = [WO.encoder_forward(WO.log_mel(c, 80), W, dims) for c in clipsd]
ref_seq = [WO.greedy_decode(e, Wg, tdims, init, rules, steps) for e in ref_encd]
ref_seq = [list(s) for s in ref_seq]
print(f"fp32 restatement: {time.time() - t0:.0f} s", file=sys.stderr)
mask = DEC.vocab_mask(tdims["n_vocab"], rules["suppress_tokens"], rules["blank_tokens"], rules["no_timestamps"])
out = {"model": "whisper-small dims, random-init weights (seeded), n_vocab 2048", "clips_encoder": 4, "clips_decisions": n_dec}
eng = pkg.ProsodyEngine(0)
for kind in ("bf16", "fp16", "fp16-resid16"):
    eng.whisper_set_operands(kind)
    eng.upload(clips4, 16000); eng.logmel_run(80)
    eng.whisper_load(dims, WW.pack(W, dims)); eng.whisper_encode_run()
    e_l2, e_max = [], []
    for i in range(4):
        got = eng.whisper_encode_fetch(i)
        e_l2.append(float(np.linalg.norm(got - ref_enc[i]) / np.linalg.norm(ref_enc[i]))); e_max.append(float(np.max(np.abs(got - ref_enc[i])) / np.std(ref_enc[i])))
    eng.whisper_decoder_load(tdims, WW.pack_decoder(Wa, tdims))
    res = eng.whisper_align(toks, frames, 3, want_cost=True)
    c_l2, same, within1, total = [], 0, 0, 0
    for i in range(4):
        cost, ti, tj = ref_al[i]
        # the restatement ran on ITS encoder output; the engine on its own: the whole chain's error
        c_l2.append(float(np.linalg.norm(res[i]["cost"] - cost) / np.linalg.norm(cost)))
        jg = res[i]["time_indices"][np.r_[True, np.diff(res[i]["text_indices"]) > 0]]
        jw = tj[np.r_[True, np.diff(ti) > 0]]
        k = min(len(jg), len(jw))
        same += int(np.sum(jg[:k] == jw[:k])); within1 += int(np.sum(np.abs(jg[:k] - jw[:k]) <= 1)); total += k
    # decisions
    eng.upload(clipsd, 16000); eng.logmel_run(80); eng.whisper_encode_run()
    eng.whisper_decoder_load(tdims, WW.pack_decoder(Wg, tdims))
    flips, n_steps, flip_margins, margins = 0, 0, [], []
    L0 = len(init)
    maxlen = max(len(s) for s in ref_seq)
    for L in range(L0, maxlen):
        act = [i for i in range(n_dec) if len(ref_seq[i]) > L]
        if not act:
            break
        prefixes = [ref_seq[i][:L] if len(ref_seq[i]) > L else ref_seq[i][:L0] for i in range(n_dec)]
        nxt, _, _ = eng.whisper_decode_step_ex(prefixes, L0, rules["eot"], rules["timestamp_begin"], mask, rules["max_initial_timestamp_index"], no_cache=True)
        for i in act:
            n_steps += 1
            if int(nxt[i]) != ref_seq[i][L]:
                flips += 1
    out[kind] = {"encoder_rel_l2_max": max(e_l2), "encoder_rel_l2_mean": float(np.mean(e_l2)), "encoder_max_err_over_std": max(e_max),
                 "cost_rel_l2_max": max(c_l2), "word_boundaries_identical": same / total, "word_boundaries_within_1_frame": within1 / total,
                 "word_boundaries": total, "greedy_steps": n_steps, "greedy_flips": flips, "greedy_flip_rate": flips / max(n_steps, 1)}
    print(kind, json.dumps(out[kind]), file=sys.stderr)
eng.close()
print(json.dumps(out))
