"""CPU tests: the C-ABI library builds, loads and exports every symbol include/pce.h declares
(no compute calls without a GPU), it refuses to run without a GPU, and the N>1 exchange step
works over gloo with world_size 2."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_and_exports_match_header():
    import __graft_entry__ as ge
    ge.build()
    from prosody_control_french_tts_amd import engine as E
    lib = ctypes.CDLL(E.native_library_path())
    header = open(os.path.join(ROOT, "include", "pce.h")).read()
    declared = set(re.findall(r"\b(pce_[a-z0-9_]+)\s*\(", header))
    assert declared == set(E.EXPORTS), declared ^ set(E.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    # ... and nothing else: the dynamic symbol table of libpce.so is exactly the header (no per-operand-type dispatch targets, no C++
    # internals, no template instantiations: -fvisibility=hidden + csrc/libpce.map)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", E.native_library_path()], stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    exported = {ln.split()[-1] for ln in nm if ln.strip()}
    assert exported == declared, sorted(exported ^ declared)
    lib.pce_api_version.restype = ctypes.c_int
    assert lib.pce_api_version() == 1 and lib.pce_api_minor() >= 3
    lib.pce_kernel_name.restype = ctypes.c_char_p
    assert [lib.pce_kernel_name(i).decode() for i in range(len(E.KERNEL_IDS))] == E.KERNEL_IDS
    # struct layouts the ctypes side assumes
    assert ctypes.sizeof(E.Slice) == 32 and ctypes.sizeof(E.Energy) == 40
    assert ctypes.sizeof(E.PitchParams) == 80 and ctypes.sizeof(E.PitchSummary) == 48


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import prosody_control_french_tts_amd as pkg
    with pytest.raises(pkg.PceError, match="no HIP device|no CPU fallback"):
        pkg.ProsodyEngine(0)


def test_product_never_imports_oracle():
    pkg_dir = os.path.join(ROOT, "prosody-control-french-tts_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
                assert "libpce_oracle" not in text, f


def test_shard_ranges_cover_in_order():
    from prosody_control_french_tts_amd import shard
    for n in (0, 1, 7, 8, 10000):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[3])
import torch.distributed as dist
from prosody_control_french_tts_amd import shard
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 11
lo, hi = shard.shard_range(n, rank, world)
full = np.arange(n * 7, dtype=np.float64).reshape(n, 7) * 0.5 - 3.0
counts = [b - a for a, b in (shard.shard_range(n, r, world) for r in range(world))]
got = shard.allgather_records(full[lo:hi], counts)          # the hot-path form: one collective, nothing but records
assert got.shape == (n, 7) and np.array_equal(got, full), got
got = shard.allgather_records(full[lo:hi])                  # sizes unknown: header row + one scalar all-reduce
assert got.shape == (n, 7) and np.array_equal(got, full), got
try:                                                        # a block that contradicts counts: flagged INSIDE the collective, every rank raises
    shard.allgather_records(full[lo:hi], [c + 1 for c in counts]); raise SystemExit("wrong counts accepted")
except shard.PeerFailure:
    pass
# a rank whose local work failed still takes part in the ONE collective and every rank learns it (no rank is left waiting)
try:
    shard.allgather_records(full[lo:hi] if rank == 0 else np.zeros((0, 7)), counts, failed=rank == 1); raise SystemExit("failure not propagated")
except shard.PeerFailure as e:
    assert "[1]" in str(e), str(e)
assert np.array_equal(shard.allgather_records(full[lo:hi], counts), full)      # ... and the group is still in step afterwards
# status barrier / agreed sections: a local exception propagates on its rank, the others raise PeerFailure, nobody hangs
assert shard.barrier(True) is True and shard.barrier(rank == 0) is False
try:
    with shard.agreed():
        if rank == 1:
            raise KeyError("local")
    raise SystemExit("rank %d left a failed section normally" % rank)
except KeyError:
    assert rank == 1
except shard.PeerFailure:
    assert rank == 0
ran = []
with shard.agreed(only_rank=0) as sec:
    if sec.mine:
        ran.append(rank)
assert ran == ([0] if rank == 0 else [])
empty = shard.allgather_records(np.zeros((0 if rank else 2, 3)))
assert empty.shape == (2, 3)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_allgather_records_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = str(29600 + os.getpid() % 300)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", ROOT, port], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


_TAGGER_WORKER = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[3]); sys.path.insert(0, os.path.join(sys.argv[3], "tests"))
import torch.distributed as dist
from prosody_control_french_tts_amd import shard, tagger as T
from test_goldens import FixtureSource, load
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
calls = []
_orig = dist.all_gather_into_tensor
def _counting(*a, **k):
    calls.append(1); return _orig(*a, **k)
dist.all_gather_into_tensor = _counting
_ar = dist.all_reduce
def _no_allreduce(*a, **k):
    raise AssertionError("the sharded tagger must not all-reduce")
dist.all_reduce = _no_allreduce
n_scen = 0
for sc in load("tagger.json"):
    n_scen += 1
    tg = T.SsmlTagger(T.ProsodySettings.from_config(sc["config"]), sc["azure_voice"], nlp=T.TablePosTagger(sc["pos_table"]))
    segs = [T.SegmentInput(s["segment"], [tuple(iv) for iv in s["intervals"]]) for s in sc["segments"]]
    res = tg.run_sharded(segs, FixtureSource(sc), rank, world, shard.allgather_records)
    for name, df in (("BDD_ssml.csv", res.bdd_ssml), ("BDD_syntagme_ssml.csv", res.bdd_syntagme_ssml),
                     ("BDD_syntagme_for_synth.csv", res.bdd_syntagme_for_synth)):
        path = os.path.join(sys.argv[5], f"r{rank}_{name}")
        df.to_csv(path, index=False)
        assert open(path, encoding="utf-8").read() == sc["expected"][name], (rank, name)
assert len(calls) == n_scen, (len(calls), n_scen)             # exactly one collective per tagger run
dist.all_reduce = _ar
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_tagger_reproduces_reference_csvs_gloo_world2(tmp_path):
    """N > 1 path of the SSML tagger: two ranks each measure half of the segments, ONE all-gather (counted below),
    and the CSVs equal the reference's (golden G7) on every rank."""
    script = tmp_path / "t.py"
    script.write_text(_TAGGER_WORKER)
    port = str(29900 + os.getpid() % 90)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", ROOT, port, str(tmp_path)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


_PIPELINE_WORKER = r'''
import os, sys, wave
import numpy as np
sys.path.insert(0, sys.argv[3]); sys.path.insert(0, os.path.join(sys.argv[3], "tests"))
import torch.distributed as dist
from pathlib import Path
from oracle import oracle as O                        # (tests may: the stub engine below answers from the CPU oracle)
from prosody_control_french_tts_amd import engine as E, shard, synth, tagger as T, textgrid_io as TG
from prosody_control_french_tts_amd.audio_pipeline import AudioPipeline
rank, world = int(sys.argv[1]), int(sys.argv[2])
base = Path(sys.argv[5])

class OracleEngine:
    """What EngineMeasurements calls on a ProsodyEngine, answered on the CPU; counts what this rank uploads."""
    uploaded = 0
    def upload(self, clips, rate):
        self.clips, self.rate, self.meter = [np.asarray(c) for c in clips], rate, 0
        OracleEngine.uploaded += len(clips)
    def lufs_set_meter_rate(self, r):
        self.meter = r
    def pitch(self, slices, params, want_f0=False):
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

def wav(path, pcm, rate):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(rate); w.writeframes(pcm.astype("<i2").tobytes())

# a voice of 5 segments (rank 0 writes it, everybody reads it)
voice = base / "Data" / "v1"; raw = base / "Data" / "v1_raw" / "audio"
if rank == 0:
    (voice / "audio").mkdir(parents=True); (voice / "WhisperTS_textgrid_files").mkdir(); raw.mkdir(parents=True)
    rng = np.random.default_rng(3)
    words = ["Bonjour", "le", "monde,", "voila", "une", "phrase.", "Tres", "longue", "ici?", "oui", "de", "la", "mer!"]
    for k in range(5):
        rate = 16000
        nat = synth.synth_clip(40 + k, seconds=1.6)
        syn = (np.roll(nat, 500).astype(np.int32) * 3 // 4).astype(np.int16)[: int(len(nat) * 0.9)]
        wav(voice / "audio" / f"segment_ph{k + 1}.wav", nat, rate); wav(raw / f"segment_ph{k + 1}.wav", syn, rate)
        t, ivs = 0.0, []
        while t < 1.3:
            d = float(np.round(rng.uniform(0.12, 0.3), 3)); ivs.append((t, t + d, str(rng.choice(words)))); t += d
            if rng.random() < 0.4:
                d = float(np.round(rng.choice([0.06, 0.16, 0.25]), 3)); ivs.append((t, t + d, " ")); t += d
        TG.write_textgrid(TG.TextGrid([TG.IntervalTier("words", ivs)], 0.0, t), voice / "WhisperTS_textgrid_files" / f"segment_ph{k + 1}.TextGrid")
cfg = {"data_dir": "Data", "out_dir": "Out", "azure_voice_name": "fr-FR-HenriNeural", "whisper_device": "cuda",
       "prosody_settings": {"baseline_window": 3, "smoothing_alpha": 0.2, "max_jump_percent": 8}, "steps_to_run": ["Measure & Build SSML"]}
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
dist.barrier()
calls = []
_orig = dist.all_gather_into_tensor
def _counting(*a, **k):
    calls.append(1); return _orig(*a, **k)
dist.all_gather_into_tensor = _counting
ap = AudioPipeline("v1", cfg, base=base, engine=OracleEngine())
res = ap.measure_prosody_and_build_ssml()                      # torch.distributed is initialised: the sharded path
assert len(calls) == 1, calls                                  # ONE collective
n_up = OracleEngine.uploaded
mine = (lambda lo_hi: lo_hi[1] - lo_hi[0])(shard.shard_range(5, rank, world))
assert n_up == 2 * mine, (n_up, mine)                          # only this rank's block was decoded and uploaded (nat + syn files); none for an empty block
sharded = {p.name: p.read_text(encoding="utf-8") for p in (ap.bdd_ssml_csv, ap.bdd_syntagme_ssml_csv, ap.bdd_syntagme_synth_csv)}
# break prediction (configs[4]): sentences sharded over the ranks, ONE more all-gather, every rank gets every label
if rank == 0:
    (voice / "transcription").mkdir(exist_ok=True)
    for k in range(5):
        (voice / "transcription" / f"segment_ph{k + 1}.txt").write_text(" ".join(["mot"] * (3 + k)) + " fin", encoding="utf-8")
dist.barrier()
class BertStub(OracleEngine):
    seen = 0
    def bert_run(self, token_lists):
        self.toks = [list(t) for t in token_lists]; BertStub.seen += len(token_lists)
    def bert_fetch(self, i):
        t = np.asarray(self.toks[i]); lab = (t % 2).astype(np.int32)           # "break after odd ids"
        return np.zeros((len(t), 2), np.float32), lab
apb = AudioPipeline("v1", cfg, base=base, engine=BertStub())
piece = lambda w: [7 + len(w)] if w == "mot" else [12, 13]
got = apb.predict_breaks(word_piecer=piece, cls_id=1, sep_id=2)
assert len(calls) == 2 and BertStub.seen == mine, (calls, BertStub.seen, mine)
assert got == {f"segment_ph{k + 1}": [0] * (3 + k) + [0] for k in range(5)}, got      # "mot" -> id 10 (even), "fin" -> first piece 12 (even)
piece2 = lambda w: [11] if w == "mot" else [12, 13]
assert apb.predict_breaks(word_piecer=piece2, cls_id=1, sep_id=2)["segment_ph3"] == [1] * 5 + [0]
dist.barrier()
if rank == 0:
    assert (apb.results_dir / "BDD_breaks.csv").read_text(encoding="utf-8").splitlines()[0] == "segment,word_index,word,break"
assert len(res.bdd_syntagme_ssml) == len(res.rows) > 5
dist.barrier(); dist.destroy_process_group()
# the same step without a process group (single rank, unsharded path) writes the same three files, text for text
cfg2 = dict(cfg, out_dir=f"Out_single_{rank}")
ap2 = AudioPipeline("v1", cfg2, base=base, engine=OracleEngine())
ap2.measure_prosody_and_build_ssml()
for p in (ap2.bdd_ssml_csv, ap2.bdd_syntagme_ssml_csv, ap2.bdd_syntagme_synth_csv):
    assert p.read_text(encoding="utf-8") == sharded[p.name], (rank, p.name)
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world", [2, 8])
def test_audio_pipeline_measure_step_shards_under_torch_distributed_gloo(tmp_path, world):
    """The PRODUCT entry point (``AudioPipeline.measure_prosody_and_build_ssml``) under an initialised process group: each rank
    decodes / uploads / measures only its block of the voice (a stub engine answering from the CPU oracle counts the uploads), ONE
    all-gather, rank 0 writes the three tables -- text for text what the single-process step writes.  World 8 = the node BASELINE config 4 names,
    on a voice of five utterances: three ranks own an EMPTY block (no upload, zero rows in the one all-gather, zero sentences for the break
    classifier) and the tables are still the single-process ones."""
    script = tmp_path / "p.py"
    script.write_text(_PIPELINE_WORKER)
    port = str(29400 + (os.getpid() + 17 * world) % 150)
    env = dict(os.environ, PCE_DIST_TIMEOUT_S="120")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), ROOT, port, str(tmp_path)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=400)[0].decode())
        except subprocess.TimeoutExpired:
            p.kill(); outs.append("no result within 400 s: " + p.communicate()[0].decode())
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, f"rank {r}:\n" + o[-3000:]


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` with no rendezvous in the environment starts two rank processes itself and rank 0
    prints the line with n_gpus = 2 (launcher + exchange on CPU: gloo, no engine)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--clips", "7", "--selftest-launcher"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["exchange_ok"] and line["records"] == 14
    # the node size the driver launches (N = 8): eight rank processes, one padded all-gather, records in rank order
    out8 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--clips", "5", "--selftest-launcher"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out8.returncode == 0, out8.stderr.decode()[-2000:]
    line8 = json.loads(out8.stdout.decode().strip().splitlines()[-1])
    assert line8["n_gpus"] == 8 and line8["exchange_ok"] and line8["records"] == 40
    # a rank whose WORLD_SIZE contradicts --gpus refuses to run instead of silently reporting n_gpus = 1
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--selftest-launcher"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert bad.returncode != 0 and b"WORLD_SIZE" in bad.stderr


def test_failure_flags_without_a_process_group():
    """The single-process forms of the rank-failure plumbing: a flagged block raises PeerFailure, ``agreed`` lets a local exception through (and a
    clean ``sys.exit(0)``), ``barrier(ok)`` returns its argument, and ``run_sharded`` re-raises the measurement error itself."""
    from prosody_control_french_tts_amd import shard, tagger as T
    assert shard.barrier(True) is True and shard.barrier(False) is False
    with pytest.raises(shard.PeerFailure):
        shard.allgather_records(np.zeros((0, 3)), [0], failed=True)
    with pytest.raises(KeyError):
        with shard.agreed():
            raise KeyError("local")
    with pytest.raises(SystemExit) as e:
        with shard.agreed(only_rank=0) as sec:
            assert sec.mine
            sys.exit(0)
    assert e.value.code == 0

    class Broken(T.MeasurementSource):
        def median_pitch(self, segment, t0=0.0, t1=None): raise OSError("device lost")
        def lufs(self, kind, segment, t0=0.0, t1=None): raise OSError("device lost")
        def duration(self, kind, segment): raise OSError("device lost")
        def part_duration(self, kind, segment, t0=0.0, t1=None): raise OSError("device lost")

    tg = T.SsmlTagger(T.ProsodySettings.from_config({}), "fr-FR-HenriNeural")
    segs = [T.SegmentInput("segment_ph1", [(0.0, 0.4, "bonjour"), (0.4, 0.6, " "), (0.6, 1.0, "monde.")])]
    with pytest.raises(OSError, match="device lost"):
        tg.run_sharded(segs, Broken(), 0, 1, shard.allgather_records)


def test_init_from_env_without_a_launcher_and_its_overrides(monkeypatch):
    """shard.init_from_env: no process group without WORLD_SIZE > 1; the device follows LOCAL_RANK unless PCE_RANK_DEVICE names it; an
    unknown PCE_DIST_BACKEND is refused before anything is initialised (the world-2 runs themselves: tests/test_gpu_world2.py on the GPU,
    the gloo workers above on CPU)."""
    import torch.distributed as dist
    from prosody_control_french_tts_amd import shard
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PCE_RANK_DEVICE", "PCE_DIST_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    assert shard.init_from_env() == (0, 1, 0) and not dist.is_initialized()
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert shard.local_device() == 3 and shard.init_from_env() == (0, 1, 3)
    monkeypatch.setenv("PCE_RANK_DEVICE", "0")
    assert shard.local_device() == 0
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "1"); monkeypatch.setenv("PCE_DIST_BACKEND", "mpi")
    with pytest.raises(ValueError):
        shard.init_from_env()
    assert not dist.is_initialized()


def test_default_engine_is_keyed_by_device():
    """engine.get_default_engine: one context per process; asking for another device than the existing engine's is an error, not a silent
    run on the wrong GPU (a stand-in engine: no GPU here)."""
    from prosody_control_french_tts_amd import engine as E

    class Stub:
        device = 2
    E.set_default_engine(Stub())
    try:
        assert E.get_default_engine() is E.get_default_engine(2)
        with pytest.raises(RuntimeError):
            E.get_default_engine(0)
    finally:
        E.set_default_engine(None)


_DEAD_PEER = r'''
import os, sys, time
rank, root, port = int(sys.argv[1]), sys.argv[2], sys.argv[3]
sys.path.insert(0, root)
os.environ.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, PCE_DIST_BACKEND="gloo",
                  PCE_RANK_DEVICE="0", PCE_DIST_TIMEOUT_S="8")
import torch
torch.cuda.set_device = lambda d: None                      # (CPU-only container: the control plane is what is under test)
import torch.distributed as dist
from prosody_control_french_tts_amd import shard
assert shard.collective_timeout_s() == 8
r, w, dev = shard.init_from_env()
assert (r, w, dev) == (rank, 2, 0)
assert shard.barrier(True) is True                         # both ranks alive: the status barrier passes
if rank == 1:
    os._exit(7)                                            # this rank dies between two collectives, without a word
t0 = time.time()
try:
    shard.barrier(True)
    print("rank 0: the barrier returned although the peer is gone"); sys.exit(3)
except Exception as e:                                     # gloo: connection reset at once, or the bounded wait
    dt = time.time() - t0
    print(f"rank 0 raised {type(e).__name__} after {dt:.1f} s")
    sys.exit(0 if dt < 30.0 else 4)
'''


def test_a_dead_peer_ends_the_surviving_rank_within_the_collective_timeout(tmp_path):
    """``shard.init_from_env`` bounds every collective (``PCE_DIST_TIMEOUT_S``): a rank whose peer has died raises inside its next status
    barrier within the bound -- the reference's pool hands back ``(ok, name)`` per voice and ends (Code/audioPipeline.py:1111-1119, 1150-1154);
    with the backend's default the survivor sat inside the barrier for 30 minutes (what took the driver's GPU run down in round 5)."""
    script = tmp_path / "dead_peer.py"
    script.write_text(_DEAD_PEER)
    port = str(29250 + os.getpid() % 140)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    procs = [subprocess.Popen([sys.executable, str(script), str(r), ROOT, port], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=120)[0].decode())
        except subprocess.TimeoutExpired:
            p.kill(); outs.append("no result within 120 s: " + p.communicate()[0].decode())
    assert procs[1].returncode == 7, outs[1][-2000:]
    assert procs[0].returncode == 0 and "rank 0 raised" in outs[0], outs[0][-3000:]
    with pytest.raises(ValueError):
        os.environ["PCE_DIST_TIMEOUT_S"] = "0"
        try:
            from prosody_control_french_tts_amd import shard
            shard.collective_timeout_s()
        finally:
            del os.environ["PCE_DIST_TIMEOUT_S"]
