#!/usr/bin/env python3
"""bench.py -- throughput of the prosody + alignment hot path on MI355X.

Default workload (BASELINE.json configs[2], "C3", the configuration the metric is quoted on): per GPU a batch of
256 synthetic 10 s 16 kHz mono clips, resident in HBM as int16 before the timed region.  One step = one pass of the
hot path over the batch:
    log-mel + Whisper-small audio encoder + teacher-forced text decoder / cross-attention DTW alignment (bf16 MFMA)
    + k_energy (gate / peak / RMS integers) + BS.1770 LUFS + Praat-AC F0 (150-600 Hz) with path finding and voiced
    median + STFT-dB 1024/256,
followed by the fetch of the per-utterance statistics (a few KB) and, for N > 1, the single all-gather of those
statistics (RCCL).  `--workload c2` (configs[1]) leaves the Whisper leg out.

    python bench.py --gpus N --steps K --warmup W
prints ONE JSON line (rank 0).  `value` = audio seconds processed by all ranks / wall time.  With --gpus N > 1 and no
WORLD_SIZE in the environment this process starts the N ranks itself (it never touches the GPU: the ranks are fresh
child processes); under `python -m torch.distributed.run ... bench.py --gpus N` it is one of the ranks.
`--gpus 8 --clips 1250` is configs[3] (C4: 10 000 clips over 8 GPUs).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (never the 2:1-sparsity figure)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # half the guide's 157.3 TFLOP/s fp32 vector rate (AMD's MI355X fp64 vector figure; the guide has no fp64 row)
VALU_F64_PEAK_TFLOPS = 78.6     # fp64 vector peak (79 TF/s)
PROFILE_DIR = os.path.join(ROOT, "profiles", "r05")
COMPOSITE = ("whisper_encoder", "whisper_align", "bert_forward", "whisper_decode_step", "whisper_decode_loop")   # brackets around several launches


def load_pmc_traffic(workload):
    """HBM bytes per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE x2 for the gfx950 wide-read
    under-count + WRITE_SIZE, KB -> B; tools/pmc_traffic.py), committed under profiles/."""
    for d in (PROFILE_DIR, os.path.join(ROOT, "profiles", "r04")):        # (the previous round's passes until this round's are committed)
        try:
            with open(os.path.join(d, f"pmc_traffic_{workload}.json")) as f:
                out = json.load(f)["bytes_per_launch"]
            out["_source"] = os.path.relpath(os.path.join(d, f"pmc_traffic_{workload}.json"), ROOT)
            return out
        except Exception:
            continue
    return None


# --------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle: test infrastructure, used here only as the thing timed BESIDE the engine)
# --------------------------------------------------------------------------------------------------------------
def _cpu_one(args):
    c, rate = args
    from oracle import oracle as O
    x = c.astype(np.float64)
    O.gate_check(c)
    O.rms_db_int16_wrapped(c)
    O.lufs_c(x, rate)
    f0 = O.pitch_ac(x / 32768.0, 1.0 / rate, 0.5 / rate, O.praat_params(150.0, 600.0))["f0"]
    v = f0[f0 > 0]
    _ = float(np.median(v)) if v.size else 0.0
    O.stft_db(c.astype(np.float32) / 32768.0)
    return len(c)


def _cpu_one_reference_shaped(args):
    """The same arithmetic called the way the reference calls it: every measurement takes a PATH and decodes the
    file again (Code/audioPipeline.py:319,327,340,360; Code/Aligners/use_whisper_timestamped.py:130,199,583)."""
    path, rate = args
    import wave
    from oracle import oracle as O

    def decode():
        with wave.open(path, "rb") as w:
            return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")

    O.gate_check(decode()); O.gate_check(decode())                  # the noise gate runs twice per file
    O.rms_db_int16_wrapped(decode())
    O.lufs_c(decode().astype(np.float64), rate)
    n = len(decode())                                                # get_duration
    f0 = O.pitch_ac(decode().astype(np.float64) / 32768.0, 1.0 / rate, 0.5 / rate, O.praat_params(150.0, 600.0))["f0"]
    v = f0[f0 > 0]
    _ = float(np.median(v)) if v.size else 0.0
    O.stft_db(decode().astype(np.float32) / 32768.0)
    return n


def cpu_prosody_baseline(clips, rate, budget_clips):
    """C2 leg of the CPU port: one thread (`value`), utterance-parallel worker processes (`all_cores`, the way the
    reference runs one process per voice, config.yaml:58) and the reference-shaped call pattern (paths in, the file
    decoded again by every measurement)."""
    sample = clips[:budget_clips]
    t0 = time.perf_counter()
    for c in sample:
        _cpu_one((c, rate))
    dt = time.perf_counter() - t0
    secs = sum(len(c) for c in sample) / rate
    out = {"value": secs / dt, "unit": "audio-seconds/sec", "cores": 1, "kind": "port", "seconds": dt, "clips": len(sample),
           "sample": f"{len(sample)} of the same synthetic 10 s clips, oracle/pce_oracle.c + numpy, {dt:.1f} s of CPU time"}
    try:
        # worker PROCESSES (fork): called before this process touches the GPU, see main()
        import multiprocessing as mp
        workers = max(1, min(len(sample), os.cpu_count() or 1, 64))
        with mp.get_context("fork").Pool(workers) as pool:
            pool.map(_cpu_one, [(sample[0], rate)] * workers)      # start-up (library load) outside the timed part
            t0 = time.perf_counter()
            pool.map(_cpu_one, [(c, rate) for c in sample], chunksize=1)
            dtp = time.perf_counter() - t0
        out["all_cores"] = {"value": secs / dtp, "processes": workers, "host_cores": os.cpu_count(), "seconds": dtp}
    except Exception as e:                                         # the single-thread figure is the contract's
        out["all_cores"] = {"error": str(e)}
    try:
        import tempfile
        import wave
        sub = sample[:32]
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for i, c in enumerate(sub):
                pth = os.path.join(td, f"clip{i}.wav")
                with wave.open(pth, "wb") as w:
                    w.setnchannels(1); w.setsampwidth(2); w.setframerate(rate); w.writeframes(c.astype("<i2").tobytes())
                paths.append(pth)
            t0 = time.perf_counter()
            for pth in paths:
                _cpu_one_reference_shaped((pth, rate))
            dtr = time.perf_counter() - t0
        out["reference_shaped"] = {"value": sum(len(c) for c in sub) / rate / dtr, "clips": len(sub), "decodes_per_file": 7, "cores": 1,
                                   "seconds": dtr}
    except Exception as e:
        out["reference_shaped"] = {"error": str(e)}
    return out


def cpu_whisper_baseline(clips, rate, model, W_enc, W_dec, dims, tdims, align_tokens, sot_len, budget_s=15.0, max_clips=4, threads=None):
    """C3 leg of the CPU port: log-mel, the audio encoder and the forced alignment of the same clips through the
    float32 restatement (oracle/whisper_oracle.py, torch CPU kernels on torch's intra-op threads), bounded by time.
    ``threads``: the intra-op threads to use -- the prosody leg's worker count, so that ONE core count describes the composite."""
    import torch
    from oracle import whisper_oracle as WO
    if threads:
        torch.set_num_threads(int(threads))
    threads = torch.get_num_threads()
    t0 = time.perf_counter()
    done = 0
    for i, c in enumerate(clips[:max_clips]):
        mel = WO.log_mel(c, dims["n_mels"])
        enc = WO.encoder_forward(mel, W_enc, dims)
        WO.find_alignment(align_tokens[i], enc, W_dec, tdims, num_frames=len(c) // 160, sot_len=sot_len)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"clips": done, "seconds": dt, "seconds_per_clip": dt / done, "threads": threads,
            "what": f"oracle/whisper_oracle.py log_mel + encoder_forward + find_alignment, Whisper-{model} size, float32"}


# --------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a rendezvous in the environment
# --------------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """Start N fresh rank processes of this script (one per GPU) and relay rank 0's JSON line.  This parent never
    imports torch.cuda nor the engine: no process that has initialised the GPU is ever re-executed."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    procs = []
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        # all ranks are watched together: once one has FAILED its peers get 20 s to leave their collectives (shard.init_from_env bounds
        # every wait, PCE_DIST_TIMEOUT_S) and are then ended -- a dead rank never leaves the launcher waiting on the others
        failed_at = None
        while any(p.poll() is None for p in procs):
            if failed_at is None and any(p.returncode not in (None, 0) for p in procs):
                failed_at = time.monotonic()
            if failed_at is not None and time.monotonic() - failed_at > 20.0:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
            time.sleep(0.1)
        rc = next((p.returncode for p in procs if p.returncode), 0)
        out0.seek(0)
        sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 6 for c3, 20 for c2)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 2 for c3, 3 for c2)")
    ap.add_argument("--clips", type=int, default=256, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--cpu-clips", type=int, default=256, help="clips timed on the CPU oracle's prosody leg (0 = no cpu_baseline)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--whisper-model", choices=["tiny", "base", "small", "medium"], default="small",
                    help="architecture of the c3 workload (BASELINE.json names small; the reference's config.yaml:15 default is medium)")
    ap.add_argument("--workload", choices=["c2", "c3"], default="c3",
                    help="c3 (default, BASELINE.json configs[2]): prosody + log-mel + Whisper encoder + forced alignment; "
                         "c2 (configs[1]): F0 + energy + LUFS + STFT only")
    ap.add_argument("--streamed-steps", type=int, default=4, help="extra steps with the batch uploaded from pinned host memory "
                    "(double buffered) for `streamed_value`; 0 = skip")
    ap.add_argument("--transcribe-steps", type=int, default=32, help="c3: free-running decoding steps per window of the extra `transcribe` "
                    "measurement (log-mel + encoder + device-resident decoding loop + forced alignment; never `value`); 0 = skip")
    ap.add_argument("--medium-steps", type=int, default=2, help="c3 with --whisper-model small on one GPU: extra steps with Whisper-medium dims (the "
                    "reference's default model) for the `medium` object; 0 = skip")
    ap.add_argument("--prosody-context", type=int, default=0, help="c3: 1 = the prosody leg (energy, LUFS, F0, STFT-dB) runs in a SECOND engine context (its own "
                    "HIP streams) on the same resident PCM, beside the Whisper leg instead of behind it")
    ap.add_argument("--framing-clips", type=int, default=1250, help="clips of the extra `framing_hbm` measurement (k_energy / k_frame_energy beyond the Infinity "
                    "Cache: the per-GPU shard of BASELINE config 4); 0 = skip")
    ap.add_argument("--dump-records", default=None, help="rank 0 writes the gathered per-utterance records of the last timed step to this .npy file")
    ap.add_argument("--first-clip", type=int, default=0, help="seed offset of the first synthetic clip (rank r owns first + [r*clips, (r+1)*clips))")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU-only check of the rank launcher and the exchange (gloo, no engine, no throughput)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 6 if args.workload == "c3" else 20
    if args.warmup is None:
        args.warmup = 2 if args.workload == "c3" else 3

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"]); rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    run_rank(args, world, rank, local_rank)


def run_rank(args, world, rank, local_rank):
    import torch
    import torch.distributed as dist
    from prosody_control_french_tts_amd import shard, synth

    rate = 16000
    n_samples = int(round(args.seconds * rate))
    counts = [args.clips] * world

    if args.selftest_launcher:
        # launcher + exchange only, on CPU: every rank contributes a record block that encodes its rank
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        rec = np.full((args.clips, 7), float(rank)); rec[:, 1] = np.arange(args.clips)
        got = shard.allgather_records(rec, counts)
        ok = got.shape == (args.clips * world, 7) and all((got[r * args.clips:(r + 1) * args.clips, 0] == r).all() for r in range(world))
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "launcher self-test (no throughput)", "value": None, "n_gpus": world, "exchange_ok": bool(ok),
                              "records": int(got.shape[0]), "data": "host-only"}))
        if not ok:
            sys.exit(1)
        return

    import prosody_control_french_tts_amd as pkg
    # synthetic data of the workload's shape; rank r owns clips [r*clips, (r+1)*clips) (weak scaling)
    clips = synth.synth_batch(args.clips, args.seconds, rate, first=args.first_clip + rank * args.clips)
    wdims = tdims = None
    if args.workload == "c3":
        from prosody_control_french_tts_amd import whisper_weights as WW
        wdims, tdims = WW.DIMS[args.whisper_model], WW.TEXT_DIMS[args.whisper_model]
        W_enc, W_dec = WW.synthetic_weights(wdims), WW.synthetic_decoder_weights(tdims)      # random-init weights of the architecture
        trng = np.random.default_rng(5 + rank)
        sot_len = 3                                    # <|sot|><|fr|><|transcribe|> ... <|eot|>: synthetic ids of a 10 s utterance's length
        align_tokens = [trng.integers(0, tdims["n_vocab"], size=int(trng.integers(24, 48))).tolist() for _ in range(args.clips)]
        align_frames = [n_samples // 160] * args.clips
    # CPU baseline first: its all-cores leg forks worker processes, which must happen before this process
    # initialises the GPU (rank 0 of a single-GPU run only: the multi-GPU lines carry "cpu_baseline": null)
    cpu = None
    if world == 1 and rank == 0 and args.cpu_clips > 0:
        cpu = cpu_prosody_baseline(clips, rate, args.cpu_clips)
        if wdims:
            wh = cpu_whisper_baseline(clips, rate, args.whisper_model, W_enc, W_dec, wdims, tdims, align_tokens, sot_len,
                                      threads=(cpu.get("all_cores", {}).get("processes") or None))
            # both legs on the SAME number of host cores, side by side (never one leg on one thread added to the other on 128, nor 64 processes beside
            # 128 threads under one "cores" figure): the prosody leg as utterance-parallel worker processes, the Whisper leg on as many torch intra-op
            # threads; the one-thread prosody rate stays beside them
            ac = cpu.get("all_cores", {})
            pros_rate = ac.get("value") or cpu["value"]
            pros_cores = ac.get("processes") or 1
            per_clip = args.seconds / pros_rate + wh["seconds_per_clip"]
            cpu["legs"] = {"prosody": {"value": pros_rate, "unit": "audio-seconds/sec", "cores": pros_cores, "what": "oracle/pce_oracle.c + numpy, one worker process per core"},
                           "prosody_one_thread": {"value": cpu["value"], "unit": "audio-seconds/sec", "cores": 1},
                           "whisper": dict(wh, value=args.seconds / wh["seconds_per_clip"], unit="audio-seconds/sec", cores=wh["threads"])}
            cpu["value"] = args.seconds / per_clip
            cpu["cores"] = pros_cores if pros_cores == wh["threads"] else {"prosody": pros_cores, "whisper": wh["threads"]}    # (one count: both legs run on the same number of cores)
            cpu["sample"] = (f"C3 per-clip time = prosody leg ({cpu['clips']} clips, C oracle, {pros_cores} worker processes: {pros_rate:.0f} x real time) + Whisper leg "
                             f"({wh['clips']} clips, {wh['seconds']:.1f} s, torch CPU float32 on {wh['threads']} threads: {args.seconds / wh['seconds_per_clip']:.2f} x real time); "
                             f"each leg uses the host's cores its own way, `legs` has them side by side; {cpu['seconds'] + wh['seconds']:.1f} s of CPU work")
    # one process per GPU over RCCL ("nccl"); PCE_DIST_BACKEND=gloo + PCE_RANK_DEVICE=0 run the same ranks on ONE device (shard.init_from_env)
    _, _, local_rank = shard.init_from_env()
    torch.cuda.set_device(local_rank)
    eng = pkg.ProsodyEngine(local_rank)
    pe = eng                                     # the context the prosody leg runs in
    if wdims and args.prosody_context:
        # two contexts on one device = two sets of HIP streams over ONE resident copy of the batch: the VALU-bound prosody kernels fill the
        # tails and the HBM-bound passes of the MFMA leg instead of queueing behind it
        pe = pkg.ProsodyEngine(local_rank)
        flat_pcm = torch.from_numpy(np.concatenate(clips)).to("cuda")
        pad = torch.zeros(64, dtype=torch.int16, device="cuda")
        pcm_dev = torch.cat([flat_pcm, pad])
        offs_dev = np.zeros(len(clips) + 1, dtype=np.int64); np.cumsum([len(c) for c in clips], out=offs_dev[1:])
        torch.cuda.synchronize()
        eng.bind_device(pcm_dev.data_ptr(), offs_dev, rate, keepalive=pcm_dev)
        pe.bind_device(pcm_dev.data_ptr(), offs_dev, rate, keepalive=pcm_dev)
        args.streamed_steps = 0
    else:
        eng.upload(clips, rate)                  # inputs resident in HBM before the timed region
    sl = pe.whole_clip_slices()
    params = pkg.PitchParams.praat(float(os.environ.get("PCE_BENCH_FLOOR", "150")), 600.0)   # reference: floor 150 (Code/audioPipeline.py:329)
    off, _ = pe.pitch_plan(sl, params)
    n_pitch_frames = int(off[1] - off[0])
    n_stft_frames = 1 + n_samples // 256
    if wdims:
        eng.whisper_load(wdims, WW.pack(W_enc, wdims))
        eng.whisper_decoder_load(tdims, WW.pack_decoder(W_dec, tdims))

    # One step = one pass of the hot path over the batch: launch() enqueues every kernel of the pass and the
    # asynchronous copy of the per-utterance statistics; finish() waits for that copy only, builds the 7-stat
    # record and all-gathers it (ONE collective).  Consecutive steps are software-pipelined (step i+1 is enqueued
    # before step i's record is read), so the device does not idle while the host unpacks: two statistic slots.
    def launch(slot):
        if wdims:
            eng.logmel_run(wdims["n_mels"])
            eng.whisper_encode_run()
            eng.whisper_align_run(align_tokens, align_frames, sot_len)   # teacher-forced decoder + cross-attention weights + DTW
            eng.whisper_align_paths_enqueue(slot)                        # every clip's (token, frame) path -> pinned host memory, asynchronously
        pe.energy_run(sl, 500)
        pe.lufs_run(sl)
        pe.pitch_run(sl, params)
        pe.stft_db_run(1024, 256)
        pe.stats_enqueue(slot)

    path_steps = []                                                # DTW path steps that reached the host per step (c3)

    def finish(slot):
        if wdims:
            pl, _, _ = eng.whisper_align_paths_wait(slot)           # what a pipeline writes TextGrids from: the alignment leaves the device inside the step
            path_steps.append(int(pl.sum()))
        r = pe.stats_wait(slot)
        en, lu, pi = r["energy"], r["lufs"][0], r["pitch"]
        # per-utterance record: [median F0, LUFS, rms, peak, silence ratio, duration, n_voiced]
        rec = np.stack([pi["median_f0"], lu, np.sqrt(en["sum_sq"] / np.maximum(en["n"], 1)), en["peak_abs"].astype(np.float64),
                        1.0 - en["n_loud"] / np.maximum(en["n"], 1), en["n"] / float(rate), pi["n_voiced"].astype(np.float64)], axis=1)
        return shard.allgather_records(rec, counts)

    done_at = []                                                   # host clock when a step's statistics had arrived (step_ms_spread)

    def run_steps(k, before_launch=None):
        rec = None
        for i in range(k):
            if before_launch:
                before_launch(i)
            launch(i & 1)
            if i > 0:
                rec = finish((i - 1) & 1); done_at.append(time.perf_counter())
        if k > 0:
            rec = finish((k - 1) & 1); done_at.append(time.perf_counter())
        return rec

    def fence():
        if shard.exchanging():
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()
        if pe is not eng:
            pe.sync()

    run_steps(args.warmup)
    if not args.no_profile:
        eng.profile_enable(True)
        eng.profile_reset()
        if pe is not eng:
            pe.profile_enable(True); pe.profile_reset()
    fence()
    done_at.clear()
    t0 = time.perf_counter()
    rec = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    # spread of the K timed steps: intervals between the arrivals of consecutive steps' statistics on the host (the steps are software
    # pipelined, so an interval is one step of device time; the first one also holds the pipeline fill).  No extra synchronisation, no extra events.
    iv = np.diff(np.array([t0] + done_at[:args.steps])) * 1e3
    spread = ({"min": float(iv[1:].min()), "median": float(np.median(iv[1:])), "max": float(iv[1:].max()), "first": float(iv[0]), "n": int(len(iv) - 1),
               "what": "ms between the arrivals of consecutive steps' statistics on the host inside the timed region (first = pipeline fill + step 1)"}
              if len(iv) > 1 else None)
    prof = eng.profile() if not args.no_profile else {}
    eng.profile_enable(False)
    if pe is not eng:
        if not args.no_profile:
            prof.update(pe.profile())
        pe.profile_enable(False)

    t_max = torch.tensor([dt], dtype=torch.float64, device="cuda" if not shard.exchanging() or dist.get_backend() == "nccl" else "cpu")
    if shard.exchanging():
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())
    audio_seconds = args.clips * args.seconds * world * args.steps
    assert rec.shape == (args.clips * world, 7)
    if args.dump_records and rank == 0:
        np.save(args.dump_records, rec)                              # the gathered per-utterance records of the last timed step (tests compare world 2 with world 1)

    # H2D-inclusive rate: the batch arrives from pinned host memory every step, double buffered (the copy of batch i+1
    # runs on a copy stream beside the kernels of batch i; the engine adopts the device buffer without a copy).
    streamed = None
    if args.streamed_steps > 0:
        try:
            flat = np.concatenate(clips)
            host = torch.from_numpy(flat).pin_memory()
            bufs = [torch.empty(flat.size + 64, dtype=torch.int16, device="cuda") for _ in range(3)]
            offs = np.zeros(len(clips) + 1, dtype=np.int64); np.cumsum([len(c) for c in clips], out=offs[1:])
            copy_stream = torch.cuda.Stream()
            ready = [torch.cuda.Event() for _ in range(3)]

            def stage(i):
                with torch.cuda.stream(copy_stream):
                    bufs[i % 3][:flat.size].copy_(host, non_blocking=True)
                    ready[i % 3].record(copy_stream)

            def before(i):
                # three buffers: the copy of batch i+1 lands in the buffer batch i-2 used, whose statistics have been
                # collected (finish(i-2) ran), while batch i-1 may still be on the device and batch i is being enqueued
                if i == 0:
                    stage(0)
                ready[i % 3].synchronize()                      # batch i is in HBM
                eng.bind_device(bufs[i % 3].data_ptr(), offs, rate, keepalive=bufs)
                if i + 1 < args.streamed_steps:
                    stage(i + 1)
            fence()
            t0 = time.perf_counter()
            run_steps(args.streamed_steps, before)
            fence()
            dts = time.perf_counter() - t0
            streamed = {"value": args.clips * args.seconds * args.streamed_steps / dts, "ms_per_step": dts / args.streamed_steps * 1e3,
                        "steps": args.streamed_steps, "h2d_bytes_per_step": int(flat.size * 2),
                        "what": "per GPU: the int16 batch is copied from pinned host memory every step on a copy stream (three device buffers) "
                                "beside the previous batch's kernels; host-side planning of the new batch included"}
        except Exception as e:                                      # never lose the main line over the extra measurement
            streamed = {"error": repr(e)}

    # What the reference's call actually runs per window (whisper_timestamped.transcribe, use_whisper_timestamped.py:150-163): log-mel +
    # encoder + FREE-RUNNING decoding + forced alignment.  Extra measurement, never `value`: the decoding loop is device-resident
    # (pce_whisper_decode_loop); end-of-text is suppressed so that every step is a live step for all clips (the worst case of a window).
    transcribe = None
    if wdims and args.transcribe_steps > 0:
        try:
            from prosody_control_french_tts_amd.Aligners import decoding as DEC
            V = tdims["n_vocab"]
            eot, ts_begin = 50257, 50364                             # the multilingual vocabulary layout
            vm = DEC.vocab_mask(V, list(range(50258, 50363)) + [eot], [220, eot], 50363)
            prompts = [[50258, 50265, 50359]] * args.clips            # <|startoftranscript|><|fr|><|transcribe|>
            N = args.transcribe_steps

            def window():
                eng.logmel_run(wdims["n_mels"]); eng.whisper_encode_run(); eng.sync()
                t0 = time.perf_counter()
                toks, _, _ = eng.whisper_decode_loop(prompts, 3, eot, ts_begin, vm, N, 50)
                t1 = time.perf_counter()
                eng.whisper_align_run(align_tokens, align_frames, sot_len); eng.sync()
                return t1 - t0, time.perf_counter() - t1, toks.shape[1]
            window()
            eng.profile_enable(False)                                 # the timed windows carry no per-kernel event pairs (70 of them per step cost 0.6 ms of one)
            fence(); tw0 = time.perf_counter()
            reps = [window() for _ in range(2)]
            fence(); tw = (time.perf_counter() - tw0) / len(reps)
            eng.profile_enable(True); eng.profile_reset()
            window()                                                  # one more window for the per-kernel split
            pr = eng.profile(); eng.profile_enable(False)
            pr = {k: dict(v, total_ms=v["total_ms"] * len(reps), launches=v["launches"] * len(reps)) for k, v in pr.items()}
            loop_s = float(np.mean([r[0] for r in reps]))
            d_, L_ = tdims["n_state"], tdims["n_layer"]
            kv_bytes = L_ * args.clips * (1500 * d_ * 2 + d_ * 1536 * 2)               # cross K rows + V^T image (key axis padded to 1536) of every layer
            # round 5: an incremental step attends from the encoder output itself (csrc/pce_xattn.inc: E once per layer), unless PCE_XATTN_ABSORB=0
            absorbed = os.environ.get("PCE_XATTN_ABSORB", "1") != "0" and d_ in (128, 256, 384, 512, 768, 1024) and tdims["n_head"] <= 16
            xkv_bytes = L_ * args.clips * 1500 * d_ * 2 if absorbed else kv_bytes
            # the loop call = cross K / V projections (once per window) + the prompt's prefix step + N - 1 incremental steps
            xproj_ms = pr.get("k_gemm_flat:xkv", {}).get("total_ms", 0.0) / len(reps)
            step0_ms = pr.get("whisper_decode_step", {}).get("total_ms", 0.0) / len(reps)
            inc_ms = (loop_s * 1e3 - step0_ms) / max(N - 1, 1)
            transcribe = {"what": f"per window of {args.clips} clips: log-mel + encoder + device-resident free-running decoding ({N} steps, end-of-text "
                                  "suppressed: every clip live at every step) + forced alignment; one upload and one download per window",
                          "window_ms": tw * 1e3, "x_real_time": args.clips * args.seconds / tw, "decode_loop_ms": loop_s * 1e3,
                          "first_step_ms_incl_cross_kv_projection": step0_ms, "cross_kv_projection_ms": xproj_ms,
                          "ms_per_incremental_step": inc_ms, "steps": int(reps[0][2]), "alignment_ms_after_loop": float(np.mean([r[1] for r in reps])) * 1e3,
                          "host_syncs_per_window": 2 + (N - 1) // 4,
                          "roofline": {"bound": "hbm", "bytes_per_step": xkv_bytes, "achieved": xkv_bytes / (inc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": xkv_bytes / (inc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "cross_attention_form": "encoder output (Q' = q Wk, U = sum p E, out = Wv U + bv)" if absorbed else "K / V^T cache",
                                       "kv_form_bytes_per_step": kv_bytes, "kv_form_equivalent_GBps": kv_bytes / (inc_ms * 1e-3) / 1e9,
                                       "kv_form_equivalent_frac": kv_bytes / (inc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "note": "algorithmic bytes of an incremental step = what its cross-attention has to read once per layer: the encoder "
                                               "output E (round 5) -- the K rows and V^T of round 3 / 4 were twice that (kv_form_bytes_per_step; "
                                               "kv_form_equivalent_GBps / _frac = those bytes / this step time, for comparison with earlier rounds' `frac`); weights "
                                               "(0.28 GB) and the self-attention cache are not counted"},
                          "kernels": {k: {"ms_per_window": v["total_ms"] / len(reps), "launches_per_window": v["launches"] / len(reps)}
                                      for k, v in pr.items() if k in ("k_cross_attn1", "k_gemm_skinny", "k_gemm_bf16", "k_attention_lean", "whisper_decode_loop",
                                                                     "whisper_encoder", "whisper_align", "k_gemm_flat:xkv")}}
        except Exception as e:                                      # never lose the main line over the extra measurement
            transcribe = {"error": repr(e)}

    # The reference's default model (config.yaml:15 `whisper_model: medium`; BASELINE.json's C3 names small): the same step with Whisper-medium
    # dims (24 + 24 layers, d = 1024, 16 heads).  Extra measurement, never `value`.  One layer's random tensors are shared by all layers (timing
    # does not depend on the values; 769 M fresh random numbers would cost more host time than the measurement).
    medium = None
    if wdims and args.whisper_model == "small" and args.medium_steps > 0 and world == 1:
        try:
            md, mt = WW.DIMS["medium"], WW.TEXT_DIMS["medium"]
            def shared(make, dims):
                one = make(dict(dims, n_layer=1))
                full = dict(one)
                for l in range(1, dims["n_layer"]):
                    for k, v in one.items():
                        if k.startswith("blocks.0."):
                            full[f"blocks.{l}." + k[len("blocks.0."):]] = v
                return full
            eng.whisper_load(md, WW.pack(shared(WW.synthetic_weights, md), md))
            eng.whisper_decoder_load(mt, WW.pack_decoder(shared(WW.synthetic_decoder_weights, mt), mt))
            m_tokens = [[int(t) % mt["n_vocab"] for t in toks] for toks in align_tokens]

            def m_step():
                eng.logmel_run(md["n_mels"]); eng.whisper_encode_run(); eng.whisper_align_run(m_tokens, align_frames, sot_len)
                pe.energy_run(sl, 500); pe.lufs_run(sl); pe.pitch_run(sl, params); pe.stft_db_run(1024, 256)
            m_step(); fence()
            eng.profile_enable(True); eng.profile_reset()
            tm0 = time.perf_counter()
            for _ in range(args.medium_steps):
                m_step()
            fence()
            tm = (time.perf_counter() - tm0) / args.medium_steps
            pr = eng.profile(); eng.profile_enable(False)
            m_flop = args.clips * (2.0 * 3000 * md["n_state"] * 240 + 2.0 * 1500 * md["n_state"] * 3 * md["n_state"]
                                   + md["n_layer"] * (2.0 * 1500 * md["n_state"] * 3 * md["n_state"] + 4.0 * 1500 * 1500 * md["n_state"]
                                                      + 2.0 * 1500 * md["n_state"] ** 2 + 16.0 * 1500 * md["n_state"] ** 2))
            enc_ms = pr.get("whisper_encoder", {}).get("total_ms", 0.0) / args.medium_steps
            gf = [(k, v) for k, v in pr.items() if k.startswith("k_gemm_flat")]
            g_ms, g_fl = sum(v["total_ms"] for _, v in gf), sum(v.get("flops", 0.0) for _, v in gf)
            medium = {"what": f"the C3 step with Whisper-medium dims (the reference's default, config.yaml:15): {args.clips} clips, {args.medium_steps} steps, "
                              "random-init weights (one layer's tensors shared by all layers)", "ms_per_step": tm * 1e3,
                      "x_real_time": args.clips * args.seconds / tm, "encoder_ms": enc_ms, "encoder_flops": m_flop,
                      "encoder_tflops": m_flop / (enc_ms * 1e-3) / 1e12 if enc_ms else None,
                      "k_gemm_flat_tflops": g_fl / (g_ms * 1e-3) / 1e12 if g_ms else None,
                      "k_gemm_flat_frac": g_fl / (g_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS if g_ms else None}
        except Exception as e:                                      # never lose the main line over the extra measurement
            medium = {"error": repr(e)}

    # north_star's ">= 70 % of the HBM roofline on the framing kernels", measured where it means HBM: the C2 / C3 batch (82 MB) sits inside the
    # 256 MB Infinity Cache, so a repeated launch on it is served on-die and launch-shaped; the rank-local shard of BASELINE config 4
    # (10 000 clips over 8 GPUs = 1 250 x 10 s = 400 MB) is beyond it.  k_energy (R3 / R7: seven exact integer reductions per sample) and
    # k_frame_energy (the VAD's 50 ms windows) over that shard, HIP events on the engine's stream.  Last measurement: it replaces the resident batch.
    framing = None
    if args.framing_clips > 0 and rank == 0:
        try:
            fclips = [clips[i % len(clips)] for i in range(args.framing_clips)]
            eng.upload(fclips, rate)
            fsl = eng.whole_clip_slices()
            fbytes = 2.0 * sum(len(c) for c in fclips)
            framing = {"what": f"{args.framing_clips} x {args.seconds:g} s clips ({fbytes / 1e6:.0f} MB of int16 PCM, the per-GPU shard of BASELINE config 4: beyond the 256 MB "
                               "Infinity Cache); algorithmic bytes = PCM in + results out; 30 launches each after 5 untimed, HIP events",
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernels": {}}
            for name, run, extra in (("k_energy", lambda: eng.energy_run(fsl, 500), 0.0),
                                     ("k_frame_energy", lambda: eng.frame_energy_run(800, 800, requantize=False), args.framing_clips * (n_samples // 800) * 12.0)):
                for _ in range(5):
                    run()
                eng.sync(); eng.profile_enable(True); eng.profile_reset()
                for _ in range(30):
                    run()
                eng.sync()
                pk = eng.profile()[name]; eng.profile_enable(False)
                ms = pk["total_ms"] / pk["launches"]
                ach = (fbytes + extra) / (ms * 1e-3) / 1e9
                framing["kernels"][name] = {"avg_launch_us": ms * 1e3, "algorithmic_bytes": fbytes + extra, "achieved": ach, "frac": ach / HBM_PEAK_GBS}
            # the framing kernels as ONE stage (as `stages` treats a stage of several kernels): their algorithmic bytes over their summed launch
            # times; the slower kernel's own fraction beside it
            tot_b = sum(k["algorithmic_bytes"] for k in framing["kernels"].values())
            tot_s = sum(k["avg_launch_us"] for k in framing["kernels"].values()) * 1e-6
            framing["achieved"] = tot_b / tot_s / 1e9
            framing["frac"] = framing["achieved"] / HBM_PEAK_GBS
            framing["min_frac"] = min(k["frac"] for k in framing["kernels"].values())
        except Exception as e:                                      # never lose the main line over the extra measurement
            framing = {"error": repr(e)}

    if rank == 0:
        # per-kernel figures (HIP events on the engine's stream around every launch)
        kernels = []
        for name, p in prof.items():
            avg_ms = p["total_ms"] / p["launches"]
            k = {"kernel": name, "avg_ms": avg_ms, "launches_per_step": p["launches"] / args.steps, "ms_per_step": p["total_ms"] / args.steps}
            if p.get("flops"):
                k["flops_per_launch"] = p["flops"] / p["launches"]
                k["achieved_tflops"] = p["flops"] / (p["total_ms"] * 1e-3) / 1e12
            if name in COMPOSITE:
                k["composite"] = True                                # a bracket around several launches, not a kernel
            kernels.append(k)
        # "k_gemm_flat:<shape>" = the launches of ONE device kernel (rocprofv3 prints k_gemm_flat<EPI>) bracketed per encoder shape:
        # they stay in `gemm_shapes`, and their sum is the `k_gemm_flat` entry the roofline is taken from
        shapes = [k for k in kernels if ":" in k["kernel"]]
        kernels = [k for k in kernels if ":" not in k["kernel"]]
        gemm_shapes = []
        if shapes:
            base = next((k for k in kernels if k["kernel"] == "k_gemm_flat"), None)
            tot_ms = sum(k["ms_per_step"] for k in shapes) + (base["ms_per_step"] if base else 0.0)
            tot_n = sum(k["launches_per_step"] for k in shapes) + (base["launches_per_step"] if base else 0.0)
            tot_fl = sum(k.get("flops_per_launch", 0.0) * k["launches_per_step"] for k in shapes) + \
                (base.get("flops_per_launch", 0.0) * base["launches_per_step"] if base else 0.0)
            if base:
                kernels.remove(base)
            kernels.append({"kernel": "k_gemm_flat", "avg_ms": tot_ms / tot_n, "launches_per_step": tot_n, "ms_per_step": tot_ms,
                            "flops_per_launch": tot_fl / tot_n, "achieved_tflops": tot_fl / (tot_ms * 1e-3) / 1e12})
            for k in shapes:
                gemm_shapes.append({"shape": k["kernel"].split(":", 1)[1], "avg_ms": k["avg_ms"], "launches_per_step": k["launches_per_step"],
                                    "ms_per_step": k["ms_per_step"], "achieved_tflops": k.get("achieved_tflops"),
                                    "frac": (k.get("achieved_tflops") or 0.0) / MFMA_BF16_PEAK_TFLOPS})
        kernels.sort(key=lambda k: -k["ms_per_step"])
        kt = {k["kernel"]: k for k in kernels}
        # stages: a stage's algorithmic bytes are moved ONCE by its kernels together; the
        # intermediates between them (candidates, autocorrelations, chunk states) are not algorithmic traffic
        pcm = 2.0 * n_samples * args.clips
        f0_out = 8.0 * n_pitch_frames * args.clips
        stft_out = 513 * 4.0 * n_stft_frames * args.clips
        stages = [
            ("energy (R3/R7)", ["k_energy"], pcm, None),
            ("lufs (R4)", ["k_lufs_pass1", "k_lufs_scan", "k_lufs_pass2", "k_lufs_gate"], pcm, None),
            ("f0 (R1: Praat AC + path + median)", ["k_pitch_frames", "k_pitch_refine", "k_pitch_delta", "k_pitch_path", "k_pitch_median"],
             pcm + f0_out, None),
            ("stft-dB (R10)", ["k_stft_raw", "k_stft_norm", "k_stft_max", "k_stft_db"], pcm + stft_out, None),     # default form: raw + norm; PCE_STFT_TWO_FFT: max + db
        ]
        if wdims:
            d, L = wdims["n_state"], wdims["n_layer"]
            flop = args.clips * (2.0 * 3000 * d * 240 + 2.0 * 1500 * d * 3 * d
                                 + L * (2.0 * 1500 * d * 3 * d + 4.0 * 1500 * 1500 * d + 2.0 * 1500 * d * d + 16.0 * 1500 * d * d))
            stages += [("log-mel (R8)", ["k_logmel_frames", "k_logmel_norm"], pcm + 80 * 3000 * 4.0 * args.clips, None),
                       (f"whisper-{args.whisper_model} encoder (R8)", ["whisper_encoder"], None, flop)]
            # (the forced-alignment leg -- decoder over 24-48 tokens per clip, alignment heads, DTW -- is timed as `whisper_align` in `kernels`)
        kernel_stage_bytes = {}
        rows = []
        for name, ks, nbytes, flops in stages:
            present = [k for k in ks if k in kt]
            if not present:
                continue
            ms = sum(kt[k]["ms_per_step"] for k in present)
            for k in present:
                kernel_stage_bytes[k] = (name, nbytes)
            dom = max(present, key=lambda k: kt[k]["ms_per_step"])
            if flops is None:
                ach = nbytes / (ms * 1e-3) / 1e9
                rows.append({"stage": name, "kernels": present, "dominant_kernel": dom, "ms_per_step": ms, "bound": "hbm",
                             "algorithmic_bytes": nbytes, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS})
                if "k_pitch_frames" in present and kt["k_pitch_frames"].get("achieved_tflops"):
                    # the frame kernel is fp64-VALU work (windowing + two real FFTs per frame, Praat's operation count), not HBM traffic:
                    # its rate against the 78.6 TFLOP/s vector fp64 peak is the meaningful fraction of this stage
                    f = kt["k_pitch_frames"]
                    rows[-1].update({"achieved_tflops_f64": f["achieved_tflops"], "peak_tflops_f64": FP64_VECTOR_PEAK_TFLOPS,
                                     "frac_f64": f["achieved_tflops"] / FP64_VECTOR_PEAK_TFLOPS,
                                     "f64_note": "k_pitch_frames only: algorithmic flops per frame (3 nw + 5 nfft log2 nfft + 1.5 nfft + maxlag) x "
                                                 "frames / its launch duration; the refinement (data-dependent Brent iterations) is not counted"})
            else:
                ach = flops / (ms * 1e-3) / 1e12
                rows.append({"stage": name, "kernels": present, "dominant_kernel": dom, "ms_per_step": ms, "bound": "mfma",
                             "algorithmic_flops": flops, "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": ach / MFMA_BF16_PEAK_TFLOPS})
        rows.sort(key=lambda r: -r["ms_per_step"])
        traffic = load_pmc_traffic(args.workload)
        # roofline of the DOMINANT KERNEL (largest device time per step among single kernels): its algorithmic work per
        # launch / its own average launch duration (HIP events on the stream it is launched on) / the peak that bounds it
        roofline = None
        leaf = [k for k in kernels if not k.get("composite")]
        if leaf:
            k0 = leaf[0]
            tr = traffic.get(k0["kernel"]) if traffic else None
            if k0.get("flops_per_launch") and k0["kernel"] != "k_pitch_frames":     # (k_pitch_frames carries fp64 VALU flops: HBM form below + its fp64 rate)
                ach = k0["flops_per_launch"] / (k0["avg_ms"] * 1e-3) / 1e12
                roofline = {"kernel": k0["kernel"], "bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": tr, "avg_launch_ms": k0["avg_ms"],
                            "launches_per_step": k0["launches_per_step"], "algorithmic_flops_per_launch": k0["flops_per_launch"],
                            "note": "dominant kernel by device time per step; achieved = 2MNK (mean over this kernel's launches in a step: "
                                    "conv, QKV / out-proj / fc1 / fc2 shapes) / its mean launch duration; dense bf16 peak 2.5 PFLOP/s at 2.4 GHz.  "
                                    "The MFMA kernels run against the power limit: the same instruction streams on all-zero operands are 14-20 % faster and the "
                                    "in-kernel clock under this load is 2.15 GHz (profiles/r04/power_probe.txt, gemm_flat_lab.txt)"}
            elif k0["kernel"] in kernel_stage_bytes:
                sname, nbytes = kernel_stage_bytes[k0["kernel"]]
                ach = nbytes / (k0["avg_ms"] * 1e-3) / 1e9
                roofline = {"kernel": k0["kernel"], "stage": sname, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach / HBM_PEAK_GBS, "traffic": tr, "avg_launch_ms": k0["avg_ms"], "algorithmic_bytes_per_launch": nbytes,
                            "note": "dominant kernel by device time per step; achieved = the algorithmic bytes of its stage (SURVEY.md 8d: "
                                    "PCM in + results out, intermediates excluded) / this kernel's own mean launch duration.  The F0 kernels are "
                                    "fp64-VALU bound (about 1e3 flop per algorithmic byte): their HBM fraction is small by construction"}
                if k0.get("achieved_tflops") and k0["kernel"] == "k_pitch_frames":
                    roofline.update({"achieved_tflops_f64": k0["achieved_tflops"], "peak_tflops_f64": FP64_VECTOR_PEAK_TFLOPS,
                                     "frac_f64": k0["achieved_tflops"] / FP64_VECTOR_PEAK_TFLOPS})
        if roofline is not None and framing and "kernels" in framing:
            # north_star's framing-kernel figure INSIDE `roofline` (the driver's record keeps `roofline`; a top-level extra key it does not)
            fk = framing["kernels"]
            roofline["framing_hbm"] = {"frac": framing["frac"], "min_frac": framing["min_frac"], "bytes": sum(k["algorithmic_bytes"] for k in fk.values()),
                                       "k_energy_GBps": fk["k_energy"]["achieved"], "k_frame_energy_GBps": fk["k_frame_energy"]["achieved"],
                                       "k_energy_frac": fk["k_energy"]["frac"], "k_frame_energy_frac": fk["k_frame_energy"]["frac"],
                                       "peak_GBps": HBM_PEAK_GBS, "clips": args.framing_clips}
        floor = None
        if wdims:
            floor_ms = flop / (MFMA_BF16_PEAK_TFLOPS * 1e12) * 1e3
            floor = {"encoder_flops": flop, "mfma_floor_ms": floor_ms, "step_frac_of_mfma_floor": floor_ms / (dt / args.steps * 1e3),
                     "note": "encoder FLOPs of the step at the dense bf16 peak / the measured step (the alignment leg's and the prosody leg's work is not in the numerator)"}
        info = eng.device_info()
        what = ("energy/gate + BS.1770 LUFS + Praat-AC F0 150-600 Hz (path finder, voiced median) + STFT-dB 1024/256"
                + (f" + log-mel + Whisper-{args.whisper_model} encoder + teacher-forced decoder / cross-attention DTW alignment "
                   f"(synthetic weights and token ids, {eng.whisper_operands} MFMA)" if wdims else ""))
        print(json.dumps({
            "metric": ("audio-seconds/sec prosody+align throughput, 16 kHz French" if wdims
                       else "audio-seconds/sec prosody throughput (no alignment leg), 16 kHz French"),
            "value": audio_seconds / dt, "unit": "audio-seconds/sec (x real-time)",
            "n_gpus": world, "dist_backend": (dist.get_backend() if shard.exchanging() else None), "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "step_ms_spread": spread,
            "timing_note": ("the timed region carries the per-kernel HIP event pairs of the profile (about 0.5 % of a step: --no-profile runs without them)"
                            if not args.no_profile else "no per-kernel events in the timed region"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": f"{eng.whisper_operands} (MFMA legs) + f64 (F0 / LUFS)" if wdims else "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload.upper()}: {args.clips} synthetic {args.seconds:g} s 16 kHz mono clips per GPU, " + what,
                       "clips_per_gpu": args.clips, "clip_seconds": args.seconds, "sample_rate": rate,
                       "parallelism": f"utterance-sharded x{world}, one all-gather of 7 fp64 stats per clip (no other collective)",
                       "prosody_context": bool(pe is not eng)},
            "roofline": roofline, "framing_hbm": framing, "alignment_path_steps_per_step": (path_steps[-1] if path_steps else None), "mfma_floor": floor, "gemm_shapes": gemm_shapes, "stages": rows, "kernels": kernels, "pmc_traffic_bytes_per_launch": traffic, "cpu_baseline": cpu,
            "streamed_value": streamed, "transcribe": transcribe, "medium": medium, "device": info["name"], "host_cores": os.cpu_count(),
        }))
    if pe is not eng:
        pe.close()
    eng.close()
    if shard.exchanging():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
