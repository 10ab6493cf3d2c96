"""BASELINE.json configs[4], the multi-rank half on CPU: ``run_all`` (Code/audioPipeline.py:1121-1166) drives the REAL steps
"Align+Transcribe" -> "Measure & Build SSML" -> "Final Transcribe" (+ ``predict_breaks``) over three voices at world size 2 under gloo.
No step is monkey-patched; what has no CPU form is stood in for below the steps: the engine (a stub answering from the CPU oracle, the
same one tests/test_abi_and_shard.py uses, plus the gate's integer sums) and the network (``transcribe_batch`` returns scripted words
derived from the clip, as tests/test_aligner_host.py does for golden G8).

Checked: the files of a voice are identical to a single-process run's, text for text (three CSVs, TextGrids, transcriptions); every
rank uploads only its block; shared folders are reset / post-processed by rank 0 only, behind status barriers (a stale file is gone, no
rank reads a half-written one); and a failure on ONE rank (its engine raises while measuring voice "bad") fails that voice on BOTH
ranks together -- no rank is left inside a collective -- and the next voice runs normally.  The GPU half of configs[4] is
tests/test_gpu_c5.py.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import json, logging, os, sys, wave
import numpy as np
sys.path.insert(0, sys.argv[3]); sys.path.insert(0, os.path.join(sys.argv[3], "tests"))
import torch.distributed as dist
from pathlib import Path
from oracle import oracle as O                        # (tests may: the stub engine below answers from the CPU oracle)
from prosody_control_french_tts_amd import audio_pipeline as AP, engine as E, shard, synth, tagger as T, textgrid_io as TG
from prosody_control_french_tts_amd.Aligners import checkpoint as CK, transcribe as TR, use_whisper_timestamped as A
from prosody_control_french_tts_amd.Aligners.tokenizer import WhisperTokenizer
rank, world = int(sys.argv[1]), int(sys.argv[2])
base = Path(sys.argv[5])
logging.basicConfig(level=logging.WARNING)

class ChainEngine:
    """The engine calls the three steps make, answered on the CPU; counts this rank's uploads per voice."""
    uploaded = 0
    fail_pitch = False
    def upload(self, clips, rate):
        self.clips, self.rate, self.meter = [np.asarray(c, dtype=np.int16) for c in clips], rate, 0
        ChainEngine.uploaded += len(clips)
    def whole_clip_slices(self):
        return list(range(len(self.clips)))
    def energy(self, slices, thr):
        out = np.zeros(len(self.clips), dtype=E.ENERGY_DTYPE)
        for i, c in enumerate(self.clips):
            x = c.astype(np.int64)
            out[i]["n"] = len(c); out[i]["sum_sq"] = int(np.sum(x * x)); out[i]["n_loud"] = int(np.sum(np.abs(c) > thr))
        return out
    def lufs_set_meter_rate(self, r):
        self.meter = r
    def pitch(self, slices, params, want_f0=False):
        if ChainEngine.fail_pitch:
            raise RuntimeError("device lost (injected on this rank)")
        out = np.zeros(len(slices), dtype=E.SUMMARY_DTYPE)
        for i, s in enumerate(slices):
            z = self.clips[s["clip"]][s["begin"]:s["end"]].astype(np.float64) / 32768.0
            try:
                f = O.pitch_ac(z, 1.0 / self.rate, float(s["x1"]), O.praat_params(150.0, 600.0))["f0"]
            except O.PraatError:
                out[i]["status"] = E.SLICE_TOO_SHORT; continue
            v = f[f > 0]
            out[i]["median_f0"] = float(np.median(v)) if v.size else 0.0
        return {"summary": out}
    def lufs(self, slices):
        vals, st = np.zeros(len(slices)), np.zeros(len(slices), dtype=np.int32)
        for i, s in enumerate(slices):
            try:
                vals[i] = O.lufs_numpy(self.clips[s["clip"]][s["begin"]:s["end"]].astype(float), self.meter or self.rate)
            except ValueError:
                st[i] = E.SLICE_TOO_SHORT
        return vals, st
    def bert_load(self, dims, blob):
        pass
    def bert_run(self, token_lists):
        self.toks = [list(t) for t in token_lists]
    def bert_fetch(self, i):
        t = np.asarray(self.toks[i]); return np.zeros((len(t), 2), np.float32), (t % 2).astype(np.int32)

WORDS = ["Bonjour", "le", "monde,", "voila", "une", "phrase.", "Tres", "longue", "ici?", "oui", "de", "la", "mer!", "encore", "un", "mot"]
def scripted_transcribe(engine, model, tokenizer, clips, opts, logger=None):
    """whisper_timestamped-shaped results that depend on the clip only (so every rank, and a single process, script the same words)."""
    out = []
    for c in clips:
        c = np.asarray(c, dtype=np.int16)
        rng = np.random.default_rng(int(np.abs(c[:4000].astype(np.int64)).sum()) % (2 ** 31))
        dur, t, words = len(c) / 16000.0, 0.05, []
        while t < dur - 0.45:
            d = float(np.round(rng.uniform(0.14, 0.32), 2))
            words.append({"text": str(rng.choice(WORDS)), "start": round(t, 2), "end": round(t + d, 2), "confidence": 0.9}); t += d
            if rng.random() < 0.35:
                t += float(rng.choice([0.06, 0.18, 0.26]))
        text = " " + " ".join(w["text"] for w in words)
        out.append({"text": text, "segments": [{"id": 0, "seek": 0, "start": words[0]["start"], "end": words[-1]["end"], "text": text, "tokens": [], "temperature": 0.0,
                                                "avg_logprob": -0.2, "compression_ratio": 1.0, "no_speech_prob": 0.0, "confidence": 0.9, "words": words}], "language": "fr"})
    return out

class NoModel:
    text_dims = {"n_vocab": 1}
    def load_into(self, engine): return self

def wav(path, pcm, rate=16000):
    path.parent.mkdir(parents=True, exist_ok=True)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(rate); w.writeframes(np.asarray(pcm, dtype="<i2").tobytes())

VOICES = {"v1": 5, "bad": 4, "v3": 3}
def lay_out(root):
    seed = 60
    for name, n in VOICES.items():
        for k in range(n):
            nat = synth.synth_clip(seed, seconds=2.2 + 0.3 * (k % 3)); seed += 1
            syn = (np.roll(nat, 400).astype(np.int32) * 3 // 4).astype(np.int16)[: int(len(nat) * 0.92)]
            wav(root / "Data" / name / "audio" / f"segment_ph{k + 1}.wav", nat); wav(root / "Data" / f"{name}_raw" / "audio" / f"segment_ph{k + 1}.wav", syn)
        wav(root / "Out" / "results" / name / "OUT.wav", synth.synth_clip(seed + 100, seconds=3.0))
        stale = root / "Data" / name / "transcription" / "stale.txt"          # a left-over of an earlier run: the step resets the folder
        stale.parent.mkdir(parents=True, exist_ok=True); stale.write_text("old")

def cfg_for(names):
    return {"data_dir": "Data", "out_dir": "Out", "azure_voice_name": "fr-FR-HenriNeural", "whisper_device": "cuda", "whisper_model": "medium",
            "voice_names": names, "multiprocessing": True, "num_processes": 5,
            "prosody_settings": {"baseline_window": 3, "smoothing_alpha": 0.2, "max_jump_percent": 8},
            "steps_to_run": ["Align+Transcribe", "Raw Synthesis", "Measure & Build SSML", "Final Transcribe"]}

def snapshot(root, name):
    out = {}
    for d in (root / "Data" / name, root / "Out" / "results" / name):
        for p in sorted(d.rglob("*")):
            if p.is_file() and p.suffix in (".csv", ".TextGrid", ".txt", ".json", ".yaml"):
                out[str(p.relative_to(root))] = p.read_text(encoding="utf-8")
    return out

A.set_model_source(model=NoModel(), tokenizer=WhisperTokenizer.toy())
CK.pad_vocab = lambda *a, **k: None
TR.transcribe_batch = scripted_transcribe            # the NETWORK is scripted; every step above it is the real one
eng = ChainEngine()
E.set_default_engine(eng)

if rank == 0:
    lay_out(base)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
dist.barrier()
gathers = []
_orig = dist.all_gather_into_tensor
def _counting(*a, **k):
    gathers.append(1); return _orig(*a, **k)
dist.all_gather_into_tensor = _counting

# the engine of rank 1 fails while measuring voice "bad": hooked on the voice's name through the pipeline's own log line
_run = AP.AudioPipeline.run
def _run_with_injection(self):
    ChainEngine.fail_pitch = (self.name == "bad" and rank == 1)
    try:
        return _run(self)
    finally:
        ChainEngine.fail_pitch = False
AP.AudioPipeline.run = _run_with_injection            # (injects the FAULT; run() itself is the product's)

failed = AP.run_all(cfg_for(list(VOICES)), base=base)
assert failed == ["bad"], (rank, failed)                                       # the SAME verdict on both ranks, nobody hung
assert len(gathers) == 3, gathers                                              # one exchange per voice's measure step ("bad" included: the flagged one)
dist.barrier()
res = base / "Out" / "results"
for name in ("v1", "v3"):
    for f in ("BDD_ssml.csv", "BDD_syntagme_ssml.csv", "BDD_syntagme_for_synth.csv", "OUT.TextGrid", "OUT.txt", "used_config.yaml"):
        assert (res / name / f).exists(), (name, f)
    assert not (base / "Data" / name / "transcription" / "stale.txt").exists()
    n = VOICES[name]
    assert sorted(p.name for p in (base / "Data" / name / "WhisperTS_textgrid_files").glob("*.TextGrid")) == sorted(f"segment_ph{k + 1}.TextGrid" for k in range(n))
    assert all((base / "Data" / name / "transcription" / f"segment_ph{k + 1}.txt").read_text(encoding="utf-8") for k in range(n))
assert not (res / "bad" / "BDD_ssml.csv").exists() and not (res / "bad" / "used_config.yaml").exists()
# per-rank work: gate + (no resample at 16 kHz) + measurements; rank r held only its block of every voice
# (per voice: the gate's two passes over the block -- the reference gates every file twice, use_whisper_timestamped.py:583 and :130 -- + the nat
# and syn files of the block for the measurements; OUT.wav of the two voices that reach "Final Transcribe" on rank 0, gated twice as well)
blocks = tuple((lambda lo_hi: lo_hi[1] - lo_hi[0])(shard.shard_range(VOICES[v], rank, world)) for v in ("v1", "bad", "v3"))   # this rank's block of v1 / bad / v3 (may be empty)
want_up = sum(4 * b for b in blocks) + (4 if rank == 0 else 0)
assert ChainEngine.uploaded == want_up, (rank, ChainEngine.uploaded, want_up)
# break prediction through the product on the voice that survived, sharded, ONE more all-gather
ap = AP.AudioPipeline("v1", cfg_for(["v1"]), base=base, engine=eng)
labels = ap.predict_breaks(word_piecer=lambda w: [7 + len(w)], cls_id=1, sep_id=2)
assert len(gathers) == 4 and list(labels) == [f"segment_ph{k + 1}" for k in range(5)]
for k, lab in labels.items():
    ws = (base / "Data" / "v1" / "transcription" / f"{k}.txt").read_text(encoding="utf-8").split()
    assert lab == [(7 + len(w)) % 2 for w in ws], (k, lab)
dist.barrier()
sharded = {name: snapshot(base, name) for name in ("v1", "v3")}
dist.barrier(); dist.destroy_process_group()
dist.all_gather_into_tensor = _orig

# ---- the same voices by ONE process (no process group): every file identical, text for text
if rank == 0:
    AP.AudioPipeline.run = _run
    solo = base / "solo"
    lay_out(solo)
    assert AP.run_all(cfg_for(["v1", "v3"]), base=solo) == []
    ap1 = AP.AudioPipeline("v1", cfg_for(["v1"]), base=solo, engine=eng)
    ap1.predict_breaks(word_piecer=lambda w: [7 + len(w)], cls_id=1, sep_id=2)
    for name in ("v1", "v3"):
        one = snapshot(solo, name)
        assert sorted(one) == sorted(sharded[name]), (sorted(set(one) ^ set(sharded[name])))
        for rel in one:
            if not rel.endswith("used_config.yaml"):                           # (it records voice_names, which differ between the two runs)
                assert one[rel] == sharded[name][rel], rel
    assert any(rel.endswith("BDD_syntagme_ssml.csv") and len(t.splitlines()) > 8 for rel, t in sharded["v1"].items())
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world", [2, 4])
def test_run_all_chain_with_a_rank_local_failure(tmp_path, world):
    """World 4: voice v3 has three utterances, so rank 3 owns an EMPTY block of it through the whole chain (aligner, measurements, the one all-gather,
    final transcription) -- what most ranks of an 8-GPU node see on a short voice."""
    script = tmp_path / "c5.py"
    script.write_text(_WORKER)
    port = str(29250 + (os.getpid() + 31 * world) % 140)
    env = dict(os.environ, PCE_DIST_TIMEOUT_S="180")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), ROOT, port, str(tmp_path)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=500)[0].decode())
        except subprocess.TimeoutExpired:
            p.kill(); outs.append("no result within 500 s: " + p.communicate()[0].decode())
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, f"rank {r}:\n" + o[-6000:]
