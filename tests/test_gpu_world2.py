"""The REAL engine under two ranks, on the one GPU a box of this pool has.

BASELINE.json configs[3] / [4] run one process per GPU over RCCL.  No multi-GPU node exists here, and RCCL refuses two ranks on one
device, so the same code path is run with ``PCE_DIST_BACKEND=gloo PCE_RANK_DEVICE=0`` (``shard.init_from_env``): two FRESH child
processes (never an exec from a process that has touched the GPU), each with its own ``ProsodyEngine`` context on cuda:0, the control
plane and the one all-gather over gloo.  Everything above the backend name is what an 8-GPU launch executes: the launcher of
``bench.py``, the rank's shard of the clips, the engine, ``shard.allgather_records``, the JSON line; ``run_all``
(Code/audioPipeline.py:1121-1166, whose parallel driver is the spawn pool of :1143-1150) with the real steps.

(a) ``bench.py --gpus 2 --workload c2 --clips 64``: ONE JSON line with ``n_gpus`` 2, and the gathered records of the 128 clips equal a
    single-rank run's of the same 128 clips bit for bit.
(b) ``run_all`` of tests/test_c5_chain.py with ``ProsodyEngine`` instead of the CPU stub: three voices cut from the reference's demo
    recordings, Align+Transcribe -> Measure & Build SSML -> Final Transcribe + ``predict_breaks``; the files of a voice are text-identical
    to a world-1 run's; a fault injected into ONE rank's engine fails that voice on BOTH ranks, the next voice runs.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


RUN_TIMEOUT_S = 240         # every child process of this file: one bound, the same treatment for a timeout and for a failed assertion
PAIR_BUDGET_S = 150         # a pair of rank processes, start to finish


def _env(**kw):
    # PCE_DIST_TIMEOUT_S: a rank whose peer has gone waits this long inside its next collective (shard.init_from_env), not gloo's 30 minutes
    env = dict(os.environ, PCE_DIST_BACKEND="gloo", PCE_RANK_DEVICE="0", PCE_DIST_TIMEOUT_S="60",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(kw)
    return env


def _tail(text, n=3000):
    if isinstance(text, bytes):
        text = text.decode(errors="replace")
    return (text or "")[-n:]


def _run(cmd, env, what):
    """One child process, bounded: a timeout fails the test the way a non-zero exit does, with the child's output."""
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=RUN_TIMEOUT_S)
    except subprocess.TimeoutExpired as e:
        pytest.fail(f"{what}: no result within {RUN_TIMEOUT_S} s\n--- stdout\n{_tail(e.stdout)}\n--- stderr\n{_tail(e.stderr)}")
    assert r.returncode == 0, f"{what}: exit {r.returncode}\n--- stdout\n{_tail(r.stdout)}\n--- stderr\n{_tail(r.stderr)}"
    return r


def _json_line(r, what):
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"{what}: {len(lines)} JSON lines\n{_tail(r.stdout)}"
    return json.loads(lines[0])


def test_bench_two_ranks_one_device_records_equal_one_rank(tmp_path):
    """ONE attempt: the gathered records of two ranks on one device equal a one-rank run's bit for bit."""
    common = ["--workload", "c2", "--steps", "3", "--warmup", "1", "--cpu-clips", "0", "--streamed-steps", "0", "--framing-clips", "0"]
    two = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--clips", "64", "--dump-records", str(tmp_path / "w2.npy")] + common,
               _env(), "bench.py --gpus 2")
    j2 = _json_line(two, "bench.py --gpus 2")
    assert j2["n_gpus"] == 2 and j2["dist_backend"] == "gloo" and j2["scaling"] == "weak" and j2["cpu_baseline"] is None
    assert j2["config"]["clips_per_gpu"] == 64 and j2["value"] > 0 and j2["roofline"] is not None
    one = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--clips", "128", "--dump-records", str(tmp_path / "w1.npy")] + common,
               _env(), "bench.py --gpus 1")
    j1 = _json_line(one, "bench.py --gpus 1")
    assert j1["n_gpus"] == 1 and j1["dist_backend"] is None
    r2, r1 = np.load(tmp_path / "w2.npy"), np.load(tmp_path / "w1.npy")
    assert r2.shape == r1.shape == (128, 7)
    assert r2.tobytes() == r1.tobytes()                     # median F0, LUFS, rms, peak, silence ratio, duration, voiced count: bit for bit
    assert (r1[:, 0] > 0).sum() > 100                       # (real measurements, not an empty table)
    # value = both ranks' audio / the slower rank's time
    assert abs(j2["value"] - 2 * 64 * 10.0 * 3 / (j2["ms_per_step"] * 3e-3)) <= 1e-6 * j2["value"]


def test_bench_eight_ranks_one_device_records_equal_one_rank(tmp_path):
    """BASELINE.json configs[3] runs EIGHT ranks with one all-gather of their statistics.  The same launcher, rank code and exchange with eight rank
    processes on the one device (gloo): one JSON line with ``n_gpus`` 8, and the gathered records of the 8 x 64 clips equal a one-rank run's of the same
    512 clips bit for bit -- eight engine contexts at once on one GPU (what the reference's worker pool does with five, config.yaml:57-58)."""
    common = ["--workload", "c2", "--steps", "2", "--warmup", "1", "--cpu-clips", "0", "--streamed-steps", "0", "--framing-clips", "0"]
    eight = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--clips", "64", "--dump-records", str(tmp_path / "w8.npy")] + common,
                 _env(), "bench.py --gpus 8")
    j8 = _json_line(eight, "bench.py --gpus 8")
    assert j8["n_gpus"] == 8 and j8["dist_backend"] == "gloo" and j8["scaling"] == "weak" and j8["config"]["clips_per_gpu"] == 64 and j8["value"] > 0
    one = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--clips", "512", "--dump-records", str(tmp_path / "w1.npy")] + common,
               _env(), "bench.py --gpus 1")
    assert _json_line(one, "bench.py --gpus 1")["n_gpus"] == 1
    r8, r1 = np.load(tmp_path / "w8.npy"), np.load(tmp_path / "w1.npy")
    assert r8.shape == r1.shape == (512, 7) and r8.tobytes() == r1.tobytes()
    assert abs(j8["value"] - 8 * 64 * 10.0 * 2 / (j8["ms_per_step"] * 2e-3)) <= 1e-6 * j8["value"]      # value = all ranks' audio / the slowest rank's time


_RCCL1 = r'''
import json, os, sys
import numpy as np
root = sys.argv[1]
sys.path.insert(0, root)
import torch, torch.distributed as dist
from prosody_control_french_tts_amd import shard
rank, world, dev = shard.init_from_env()
assert (rank, world, dev) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == "nccl" and shard.exchanging()
assert shard.barrier(True) is True and shard.barrier(False) is False                  # the status all-reduce, on a device tensor
rec = np.arange(21, dtype=np.float64).reshape(3, 7) / 7.0
got = shard.allgather_records(rec, [3])                                               # ONE all_gather_into_tensor of device memory
assert got.tobytes() == rec.tobytes()
assert shard.allgather_records(rec).tobytes() == rec.tobytes()                        # sizes agreed first (MAX all-reduce + header row)
assert shard.allgather_records(np.zeros((0, 7)), [0]).shape == (0, 7)
for bad in (dict(failed=True), dict(counts=[4])):
    try:
        shard.allgather_records(rec, bad.get("counts", [3]), failed=bad.get("failed", False)); raise SystemExit("no PeerFailure")
    except shard.PeerFailure:
        pass
try:
    with shard.agreed():
        raise KeyError("local")
except KeyError:
    pass
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL1 OK", torch.cuda.get_device_name(0))
'''


def test_rccl_code_path_runs_with_one_rank(tmp_path):
    """What a one-GPU box can run of the RCCL form: the "nccl" backend refuses two ranks on one device, so ``PCE_DIST_WORLD1=1`` makes
    the launcher's ONE rank join a process group and the collectives of ``shard`` really execute (device tensors through RCCL: communicator
    set-up with ``device_id``, the status all-reduce, the padded ``all_gather_into_tensor`` with its failure row) instead of taking the
    no-group shortcut; then ``bench.py`` the same way: one JSON line that says ``dist_backend: "nccl"``, records bit-equal to the run
    without a process group.  More than one rank over RCCL / xGMI stays unmeasured on hardware."""
    env = _env(PCE_DIST_BACKEND="nccl", PCE_DIST_WORLD1="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29600 + os.getpid() % 200))
    script = tmp_path / "rccl1.py"
    script.write_text(_RCCL1)
    r = _run([sys.executable, str(script), ROOT], env, "one-rank RCCL group")
    assert "RCCL1 OK" in r.stdout, _tail(r.stdout) + _tail(r.stderr)
    common = ["--workload", "c2", "--steps", "3", "--warmup", "1", "--cpu-clips", "0", "--streamed-steps", "0", "--clips", "64", "--framing-clips", "0"]
    # the driver's own launch line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    # --gpus N ...`) with N = 1: the launcher sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    a = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
              "--master-port", str(29800 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "1",
              "--dump-records", str(tmp_path / "rccl.npy")] + common, _env(PCE_DIST_BACKEND="nccl", PCE_DIST_WORLD1="1"), "bench.py under torch.distributed.run")
    ja = _json_line(a, "bench.py under torch.distributed.run")
    assert ja["n_gpus"] == 1 and ja["dist_backend"] == "nccl" and ja["value"] > 0
    b = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dump-records", str(tmp_path / "plain.npy")] + common, _env(), "bench.py")
    assert _json_line(b, "bench.py")["dist_backend"] is None
    assert np.load(tmp_path / "rccl.npy").tobytes() == np.load(tmp_path / "plain.npy").tobytes()


_WORKER = r'''
import logging, os, sys
import numpy as np
rank, world, root, port, base = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
sys.path.insert(0, root)
os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                  PCE_DIST_BACKEND="gloo", PCE_RANK_DEVICE="0")
from pathlib import Path
import torch.distributed as dist
from prosody_control_french_tts_amd import audio_pipeline as AP, bert_weights as BW, engine as E, shard, tagger as T
from tests.test_gpu_c5 import WORDS, raw_synthesis_stand_in, write_wav
from tests.test_gpu_aligner import write_model_dir
base = Path(base)
logging.basicConfig(level=logging.WARNING)

# three voices cut from the reference's ten demo recordings (44.1 kHz, 2.9-37.2 s)
z = np.load(Path(root) / "tests" / "golden" / "demo_full.npz")
rate = int(z["rate"])
names = sorted((k for k in z.files if k != "rate"), key=T.segment_sort_key)
VOICES = {"v1": names[:5], "bad": names[6:10], "v3": names[4:7]}            # (v3 holds the 37.2 s recording: two Whisper windows)

def lay_out(b):
    for voice, segs in VOICES.items():
        for k, n in enumerate(segs):
            nat = z[n]
            write_wav(b / "Data" / "voice" / voice / "audio" / f"segment_ph{k + 1}.wav", nat, rate)
            write_wav(b / "Data" / "voice" / f"{voice}_raw" / "audio" / f"segment_ph{k + 1}.wav", raw_synthesis_stand_in(nat, k), 16000)
        out = np.concatenate([raw_synthesis_stand_in(z[n], k) for k, n in enumerate(segs[:2])])[: 20 * 16000]
        write_wav(b / "Out" / "results" / voice / "OUT.wav", out, 16000)

def cfg_for(voices, model_root):
    return {"data_dir": "Data/voice", "out_dir": "Out", "azure_voice_name": "fr-FR-HenriNeural", "whisper_device": "cuda", "whisper_model": "medium",
            "whisper_dir": str(model_root), "voice_names": voices, "multiprocessing": True, "num_processes": 5,
            "prosody_settings": {"baseline_window": 3, "smoothing_alpha": 0.2, "max_jump_percent": 8},
            "steps_to_run": ["Align+Transcribe", "Raw Synthesis", "Measure & Build SSML", "Final Transcribe"]}

def snapshot(b, voice):
    out = {}
    for d in (b / "Data" / "voice" / voice, b / "Out" / "results" / voice):
        for p in sorted(d.rglob("*")):
            if p.is_file() and p.suffix in (".csv", ".TextGrid", ".txt", ".json"):
                out[str(p.relative_to(b))] = p.read_text(encoding="utf-8")
    return out

model_root = base / "whisper_dir"
if rank == 0:
    lay_out(base)
    write_model_dir(model_root, merges=WORDS, word_gain=3.0, eot_gain=0.3)
r, w, dev = shard.init_from_env()
assert (r, w, dev) == (rank, world, 0) and dist.get_backend() == "gloo"
dist.barrier()

# the product's own engine (no stub): created lazily by the first step; count this rank's uploads, inject the fault below the steps
uploads = []
_upload = E.ProsodyEngine.upload
def _counting_upload(self, clips, rate_):
    uploads.append(len(clips)); return _upload(self, clips, rate_)
E.ProsodyEngine.upload = _counting_upload
fail = {"on": False}
_pitch = E.ProsodyEngine.pitch
def _failing_pitch(self, *a, **k):
    if fail["on"]:
        raise RuntimeError("device lost (injected on this rank)")
    return _pitch(self, *a, **k)
E.ProsodyEngine.pitch = _failing_pitch
_run = AP.AudioPipeline.run
def _run_with_injection(self):
    fail["on"] = (self.name == "bad" and rank == 1)
    try:
        return _run(self)
    finally:
        fail["on"] = False
AP.AudioPipeline.run = _run_with_injection
gathers = []
_orig = dist.all_gather_into_tensor
def _counting(*a, **k):
    gathers.append(1); return _orig(*a, **k)
dist.all_gather_into_tensor = _counting

failed = AP.run_all(cfg_for(list(VOICES), model_root), base=base)
assert failed == ["bad"], (rank, failed)                                   # the SAME verdict on both ranks, nobody hung in a collective
assert len(gathers) == 3, gathers                                          # one exchange per voice's measure step
eng = E.get_default_engine()
assert type(eng) is E.ProsodyEngine and eng.device == 0 and eng.device_info()["compute_units"] == 256
n_mine = {v: (lambda lo_hi: lo_hi[1] - lo_hi[0])(shard.shard_range(len(s), rank, world)) for v, s in VOICES.items()}
assert uploads and all(u <= 2 * max(n_mine.values()) for u in uploads), uploads          # this rank's blocks (nat + syn), never a whole voice
dist.barrier()
res = base / "Out" / "results"
for v in ("v1", "v3"):
    for f in ("BDD_ssml.csv", "BDD_syntagme_ssml.csv", "BDD_syntagme_for_synth.csv", "OUT.TextGrid", "OUT.txt", "used_config.yaml"):
        assert (res / v / f).exists(), (v, f)
    assert sorted(p.name for p in (base / "Data" / "voice" / v / "WhisperTS_textgrid_files").glob("*.TextGrid")) == \
        sorted(f"segment_ph{k + 1}.TextGrid" for k in range(len(VOICES[v])))
assert not (res / "bad" / "BDD_ssml.csv").exists()
dims = dict(BW.DIMS["tiny"], n_pos=128)
piece = lambda wd: [3 + (sum(map(ord, wd)) % 250), 5 + len(wd) % 200][: 1 + len(wd) % 2]
ap = AP.AudioPipeline("v1", cfg_for(["v1"], model_root), base=base)
labels = ap.predict_breaks(word_piecer=piece, weights=BW.synthetic_weights(dims, seed=5), dims=dims, cls_id=1, sep_id=2)
assert len(gathers) == 4 and list(labels) == [f"segment_ph{k + 1}" for k in range(5)]
dist.barrier()
sharded = {v: snapshot(base, v) for v in ("v1", "v3")}
sharded_labels = {k: list(v) for k, v in labels.items()}
dist.barrier(); dist.destroy_process_group()
dist.all_gather_into_tensor = _orig

# ---- the same voices by ONE process (no process group) on the same engine: every file identical, text for text
if rank == 0:
    AP.AudioPipeline.run = _run
    solo = base / "solo"
    lay_out(solo)
    assert AP.run_all(cfg_for(["v1", "v3"], model_root), base=solo) == []
    ap1 = AP.AudioPipeline("v1", cfg_for(["v1"], model_root), base=solo)
    l1 = ap1.predict_breaks(word_piecer=piece, weights=BW.synthetic_weights(dims, seed=5), dims=dims, cls_id=1, sep_id=2)
    assert {k: list(v) for k, v in l1.items()} == sharded_labels
    n_rows = 0
    for v in ("v1", "v3"):
        one = snapshot(solo, v)
        assert sorted(one) == sorted(sharded[v]), sorted(set(one) ^ set(sharded[v]))
        import hashlib
        print("raw-json hashes", v, {rel.rsplit("/", 1)[1]: (hashlib.sha1(one[rel].encode()).hexdigest()[:10], hashlib.sha1(sharded[v][rel].encode()).hexdigest()[:10])
                                    for rel in one if rel.endswith(".raw.json")})
        for rel in one:
            if one[rel] != sharded[v][rel]:
                import difflib
                print("\n".join(list(difflib.unified_diff(one[rel].splitlines(), sharded[v][rel].splitlines(), "world1", "world2", lineterm="", n=1))[:60]))
            assert one[rel] == sharded[v][rel], rel
        n_rows += len(one[f"Out/results/{v}/BDD_syntagme_ssml.csv"].splitlines()) - 1
    assert n_rows >= 8, n_rows
eng.close()
print("rank", rank, "ok")
'''


def _run_pair(base):
    """Two fresh rank processes, watched TOGETHER: the moment one exits non-zero its peer gets 10 s to fail on its own (PeerFailure, or the
    bounded wait of shard.init_from_env) and is then ended; the pair as a whole has PAIR_BUDGET_S.  -> (ok, [output of rank 0, of rank 1])"""
    import time
    base.mkdir()
    script = base / "w2.py"
    script.write_text(_WORKER)
    port = str(29400 + os.getpid() % 140)
    logs = [open(base / f"rank{r}.log", "w+") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", ROOT, port, str(base)], stdout=logs[r], stderr=subprocess.STDOUT, env=_env())
             for r in range(2)]
    t0 = time.monotonic()
    failed_at, verdict = None, ""
    while any(p.poll() is None for p in procs):
        now = time.monotonic()
        if failed_at is None and any(p.returncode not in (None, 0) for p in procs):
            failed_at = now
        if (failed_at is not None and now - failed_at > 10.0) or now - t0 > PAIR_BUDGET_S:
            verdict = f"ended by the test after {now - t0:.0f} s ({'peer failed' if failed_at is not None else 'budget of %d s spent' % PAIR_BUDGET_S})"
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.1)
    outs = []
    for f in logs:
        f.seek(0); outs.append(f.read()); f.close()
    ok = all(p.returncode == 0 and f"rank {r} ok" in o for r, (p, o) in enumerate(zip(procs, outs)))
    if not ok:
        outs = [f"[exit {p.returncode}{'; ' + verdict if verdict else ''}]\n{o}" for p, o in zip(procs, outs)]
    return ok, outs


def test_run_all_with_the_real_engine_at_world2_and_a_rank_local_failure(tmp_path):
    """ONE attempt, and the comparison with the world-1 run is EXACT (text for text).  On failure both ranks' output is in the report."""
    ok, outs = _run_pair(tmp_path / "pair")
    assert ok, "\n".join(f"===== rank {r} (last 3000 characters)\n{_tail(o)}" for r, o in enumerate(outs))


_ALIGN_AB = r'''
import json, logging, os, sys
import numpy as np
root, base = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
from pathlib import Path
from prosody_control_french_tts_amd import tagger as T
from prosody_control_french_tts_amd.Aligners import use_whisper_timestamped as A
from tests.test_gpu_c5 import WORDS, write_wav
from tests.test_gpu_aligner import write_model_dir
base = Path(base)
logging.basicConfig(level=logging.ERROR)
z = np.load(Path(root) / "tests" / "golden" / "demo_full.npz")
rate = int(z["rate"])
names = sorted((k for k in z.files if k != "rate"), key=T.segment_sort_key)
for k, n in enumerate(names):
    write_wav(base / "audio" / f"segment_ph{k + 1}.wav", z[n], rate)
write_model_dir(base / "whisper_dir", merges=WORDS, word_gain=3.0, eot_gain=0.3)
os.environ["PCE_WHISPER_DIR"] = str(base / "whisper_dir")
A.main(str(base / "audio"), str(base / "out"), whisper_model="medium", device="cuda")
print("ALIGN OK")
'''


def test_aligner_files_do_not_depend_on_the_decoding_kernels(tmp_path):
    """The "Align+Transcribe" step over the reference's ten demo recordings (2.9-37.2 s: two Whisper windows, VAD, free-running decoding, forced
    alignment) in two fresh processes: the round-5 decoding kernels (cross-attention from the encoder output, row-major self-attention caches)
    against the round-3 / 4 ones (``PCE_XATTN_ABSORB=0 PCE_SELF_ROWS=0``).  Every TextGrid and transcription is text-identical; the raw JSON
    holds the same words with the same times, confidences within 0.002 (they are probabilities of the same tokens computed through different
    sums)."""
    outs = {}
    for tag, extra in (("new", {}), ("old", {"PCE_XATTN_ABSORB": "0", "PCE_SELF_ROWS": "0"})):
        base = tmp_path / tag
        base.mkdir()
        script = base / "run.py"
        script.write_text(_ALIGN_AB)
        env = _env(**extra)
        env.pop("PCE_DIST_BACKEND", None); env.pop("PCE_RANK_DEVICE", None)
        r = _run([sys.executable, str(script), ROOT, str(base)], env, f"Align+Transcribe, {tag} kernels")
        assert "ALIGN OK" in r.stdout, _tail(r.stdout) + _tail(r.stderr)
        outs[tag] = base
    def files(b, sub, suffix):
        return sorted(p for p in (b / sub).glob("*" + suffix))
    tg_new, tg_old = files(outs["new"], "out", ".TextGrid"), files(outs["old"], "out", ".TextGrid")
    assert len(tg_new) == 10 and [p.name for p in tg_new] == [p.name for p in tg_old]
    for a, b in zip(tg_new, tg_old):
        assert a.read_text(encoding="utf-8") == b.read_text(encoding="utf-8"), a.name
    for a, b in zip(files(outs["new"], "out_transcription", ".txt"), files(outs["old"], "out_transcription", ".txt")):
        assert a.read_text(encoding="utf-8") == b.read_text(encoding="utf-8"), a.name
    n_words = 0
    for a, b in zip(files(outs["new"], "out_raw_json", ".raw.json"), files(outs["old"], "out_raw_json", ".raw.json")):
        ja, jb = json.loads(a.read_text(encoding="utf-8")), json.loads(b.read_text(encoding="utf-8"))
        assert ja["text"] == jb["text"] and len(ja["segments"]) == len(jb["segments"]), a.name
        for sa, sb in zip(ja["segments"], jb["segments"]):
            assert sa["tokens"] == sb["tokens"] and (sa["start"], sa["end"]) == (sb["start"], sb["end"]), a.name
            assert [(w["text"], w["start"], w["end"]) for w in sa["words"]] == [(w["text"], w["start"], w["end"]) for w in sb["words"]], a.name
            for wa, wb in zip(sa["words"], sb["words"]):
                assert abs(wa["confidence"] - wb["confidence"]) <= 0.002, (a.name, wa, wb)
                n_words += 1
    assert n_words > 50

