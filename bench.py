#!/usr/bin/env python3
"""Benchmark of the FunCodec encode+decode hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one Speech2Token(run_mod="inference") pass (encode -> 32-stage RVQ -> decode) over one batch
of 16 synthetic 10 s / 16 kHz utterances PER GPU on the 16k-nq32ds640 architecture (BASELINE.json
configs[1]); for N > 1 the utterances are sharded across ranks (weak scaling: 16 per rank) and the code
indices are all-gathered over RCCL inside the timed step.  Inputs are resident in HBM when timing starts.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

UTTS_PER_GPU = 16
SAMPLES = 160000
CONFIG = os.environ.get("FC_BENCH_CONFIG", "ds640")   # the contract metric is ds640; other recipes only for side measurements
PEAK_F32_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: fp32 matrix (= vector) peak
PEAK_HBM_TBS = 8.0


def cpu_baseline(sample_utts: int = 4):
    """The oracle (ATen-CPU restatement of the reference path, pinned bit-exact against the real
    reference in the build container) timed on this host's cores on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from torch_oracle import Oracle
    from funcodec_amd.config import arch_from_config, recipe_config
    from funcodec_amd.synth import make_state_dict, synthetic_audio
    cfg = recipe_config(CONFIG)
    sd = make_state_dict(arch_from_config(cfg), 0)
    orc = Oracle(cfg, {k: torch.from_numpy(v) for k, v in sd.items()})
    default_threads = torch.get_num_threads()
    x = torch.from_numpy(synthetic_audio(sample_utts, SAMPLES, 1234))
    orc.inference(x[:1, :16000])                      # warm-up (thread pools, LSTM weight flatten)
    runs = {}
    for threads in sorted({default_threads, min(default_threads, 32)}, reverse=True):
        torch.set_num_threads(threads)
        orc.inference(x[:1, :16000])
        t0 = time.perf_counter()
        orc.inference(x)
        runs[threads] = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    best = min(runs, key=runs.get)                   # give the CPU path its better thread count
    return {"value": round(sample_utts * SAMPLES / 16000.0 / runs[best], 3), "unit": "audio-s/s", "cores": best,
            "kind": "port", "seconds": round(sum(runs.values()), 2),
            "by_threads": {str(k): round(sample_utts * SAMPLES / 16000.0 / v, 3) for k, v in runs.items()},
            "sample": f"oracle/torch_oracle.py (same ATen CPU kernels as the reference's PyTorch path), "
                      f"{sample_utts} x 10 s utterances of the benchmark batch, ds640, n_q=32, one timed run per thread count "
                      f"after a 1 s warm-up; best thread count reported"}


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed PMC passes (tools/collect_profiles.sh: FETCH_SIZE and WRITE_SIZE in
    separate rocprofv3 --pmc runs of this same benchmark).  Correction per MI355X_MICROARCH.md: FETCH_SIZE x 2 on gfx950
    (checked here on the 32->32 k=1 convs whose byte count is known: 0.320 GB reported for 0.656 GB read), WRITE_SIZE x 1
    (0.641 GB reported for 0.656 GB written).  None if the committed profile has no entry for this instantiation."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_hbm_traffic_pmc.json")))
    if not files:
        return None
    try:
        for k in json.load(open(files[-1]))["per_kernel"]:
            if k["kernel"] == kernel and k["launches"]:
                return {"bytes_per_launch": round((2.0 * k["fetch_kb"] + k["write_kb"]) * 1024.0 / k["launches"]),
                        "fetch_x2_gb": round(2.0 * k["fetch_kb"] * 1024.0 / 1e9, 3), "write_gb": round(k["write_kb"] * 1024.0 / 1e9, 3),
                        "launches": k["launches"], "source": os.path.basename(files[-1])}
    except (OSError, KeyError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-event-profile", action="store_true",
                    help="do not bracket kernels with HIP events inside the timed region")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from funcodec_amd.config import arch_from_config, recipe_config
    from funcodec_amd.model import EncodecMI355X
    from funcodec_amd.parallel import gather_codes, shard_range
    from funcodec_amd.synth import make_state_dict, synthetic_audio

    cfg = recipe_config(CONFIG)
    arch = arch_from_config(cfg)
    model = EncodecMI355X(arch, f"cuda:{local_rank}")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(arch, 0).items()})
    eng = model.engine

    # global batch = 16 utterances per GPU; each rank takes its contiguous slice
    total_utts = UTTS_PER_GPU * world
    lo, hi = shard_range(total_utts, rank, world)
    shard_sizes = [shard_range(total_utts, r, world)[1] - shard_range(total_utts, r, world)[0] for r in range(world)]
    wav_all = synthetic_audio(total_utts, SAMPLES, 1234) if total_utts <= 32 else None
    if wav_all is None:   # avoid generating 1 GB of noise per rank at large N: per-rank seeds
        wav = torch.from_numpy(synthetic_audio(hi - lo, SAMPLES, 1234 + rank)).cuda()
    else:
        wav = torch.from_numpy(wav_all[lo:hi]).cuda()
    n_q = arch.num_quantizers

    def step():
        r = eng.encode_decode(wav, n_q, use_scale=True)
        codes = gather_codes(r["codes"], dist, shard_sizes=shard_sizes) if world > 1 else r["codes"]
        return r, codes

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.set_profiling(not args.no_event_profile)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r, codes = step()
    fence()
    dt = time.perf_counter() - t0
    prof = eng.read_profile() if not args.no_event_profile else []
    eng.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(r["recon"]).all())
    assert codes.shape[1] == total_utts

    if rank == 0:
        audio_s = total_utts * SAMPLES / 16000.0 * args.steps
        work = eng.work(UTTS_PER_GPU, SAMPLES, n_q)
        out = {
            "metric": "audio-seconds encoded+decoded per wall-sec, 16k-nq32ds640" if CONFIG == "ds640" else
                      f"audio-seconds encoded+decoded per wall-sec, recipe {CONFIG} (side measurement, not the contract metric)",
            "value": round(audio_s / dt, 2), "unit": "audio-s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "encodec 16k-nq32ds640 (57.6M, synthetic seeded checkpoint), run_mod=inference "
                                   "(encode + 32-stage RVQ + decode), 16 x 10 s utterances per GPU, n_q=32",
                       "utterances_per_gpu": UTTS_PER_GPU, "samples_per_utterance": SAMPLES,
                       "global_utterances": total_utts,
                       "parallelism": f"utterance-sharded x{world}, all_gather(codes) over RCCL" if world > 1 else "single GPU"},
            "algorithmic_per_step_per_gpu": {"tflop": round(work["total_flops"] / 1e12, 4),
                                             "conv_tflop": round(work["conv_flops"] / 1e12, 4),
                                             "conv_gb": round(work["conv_bytes"] / 1e9, 3),
                                             "launches": work["total_launches"]},
        }
        if prof:
            kern = []
            for p in prof:
                if p["launches"] == 0:
                    continue
                ms = p["total_ms"]
                kern.append({"kernel": p["kernel"], "launches_per_step": p["launches"] // args.steps,
                             "ms_per_step": round(ms / args.steps, 3),
                             "avg_us_per_launch": round(ms * 1e3 / p["launches"], 2),
                             "tflops": round(p["flops"] / (ms * 1e-3) / 1e12, 2) if ms > 0 else None,
                             "alg_gbs": round(p["bytes"] / (ms * 1e-3) / 1e9, 1) if ms > 0 and p["bytes"] else None})
            dom = max((k for k in kern if k["kernel"].startswith("conv_")), key=lambda k: k["ms_per_step"])
            conv_ms = sum(k["ms_per_step"] for k in kern if k["kernel"].startswith("conv_"))
            conv_fl = sum(p["flops"] for p in prof if p["kernel"].startswith("conv_")) / args.steps
            out["roofline"] = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"],
                               "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(dom["tflops"] / PEAK_F32_TFLOPS, 4), "traffic": (pmc_traffic(dom["kernel"]) or {}).get("bytes_per_launch"),
                               "traffic_detail": pmc_traffic(dom["kernel"]),
                               "algorithmic_bytes_per_launch": round(dom["alg_gbs"] * 1e9 * dom["avg_us_per_launch"] * 1e-6) if dom["alg_gbs"] else None,
                               "avg_us_per_launch": dom["avg_us_per_launch"],
                               "launches_per_step": dom["launches_per_step"],
                               "hbm_alg_gbs": dom["alg_gbs"],
                               "all_conv_instantiations": {"ms_per_step": round(conv_ms, 3),
                                                           "tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                                                           "frac": round(conv_fl / (conv_ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS, 4)},
                               "note": "fp32-input MFMA (exact fp32, peak = fp32 vector peak); achieved = algorithmic "
                                       "FLOPs of this kernel's launches / sum of their HIP-event durations in the timed region"}
            out["kernels"] = kern
            out["whole_step"] = {"tflops": round(work["total_flops"] / (dt / args.steps) / 1e12, 2),
                                 "frac_of_f32_peak": round(work["total_flops"] / (dt / args.steps) / 1e12 / PEAK_F32_TFLOPS, 4),
                                 "alg_hbm_tbs": round(work["total_bytes"] / (dt / args.steps) / 1e12, 3),
                                 "frac_of_hbm_peak": round(work["total_bytes"] / (dt / args.steps) / 1e12 / PEAK_HBM_TBS, 4)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
