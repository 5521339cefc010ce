"""`python bench.py --gpus N` must launch its own N ranks (the reference's multi-GPU story is N self-launched
processes, egs/LibriTTS/codec/encoding_decoding.sh:59-101).  Runs the launcher's dry-run mode on CPU: world 2 over gloo, the same
shard_range / gather_codes the GPU ranks use, no engine."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=300)


def _json_lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{")]


def test_bench_self_launches_its_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run-launcher"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["config"]["ranks_seen"] == 2 and out["gather_ok"] is True
    assert out["config"]["shard_sizes"] == [4, 3] and out["config"]["global_utterances"] == 7


def test_bench_single_rank_dry_run_needs_no_launcher():
    r = _run(["--gpus", "1", "--steps", "1", "--dry-run-launcher"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_lines(r.stdout)[0]
    assert out["n_gpus"] == 1 and out["config"]["ranks_seen"] == 1 and out["gather_ok"] is True


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _run(["--gpus", "4", "--dry-run-launcher"], env_extra={"WORLD_SIZE": "1", "RANK": "0"}, drop=())
    # under an existing WORLD_SIZE the script must not re-launch, and a rank count that contradicts --gpus is an error, not a silent N = 1 run
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr and not _json_lines(r.stdout)


def test_roofline_objects_carry_both_fractions_and_the_traffic_ratio():
    """the bench line reports the fraction of the roof at the SUSTAINED clock next to the nominal one, the measured / algorithmic
    traffic ratio, and the dominant conv class inside the `roofline` object the driver parses.  Pure host logic over a synthetic profile."""
    sys.path.insert(0, ROOT)
    import bench
    prof = [dict(kernel="lstm_persist_kernel<NS>", total_ms=12.6, flops=100.66e9 * 10, bytes=5e9, launches=10),
            dict(kernel="conv_mfma_kernel<128, 128, 2, 2, 0, 9, false>", total_ms=11.3, flops=1.2e12, bytes=267e6 * 15, launches=15),
            dict(kernel="reshead_kernel<32>", total_ms=2.0, flops=1e9, bytes=8e9, launches=10)]
    out = bench.kernel_rooflines(prof, 5)
    top, conv, hbm = out["roofline"], out["roofline_conv"], out["roofline_hbm"]
    assert top["kernel"].startswith("lstm_persist_kernel") and conv["kernel"].startswith("conv_mfma_kernel")
    for o in (top, conv):
        assert 0 < o["frac"] < o["frac_at_sustained_clock"] < 1 and o["sustained_clock_ghz"] == bench.SUSTAINED_GHZ
        assert abs(o["frac_at_sustained_clock"] * bench.PEAK_F32_SUSTAINED - o["achieved"]) < 0.05
    assert top["conv_class"]["kernel"] == conv["kernel"] and top["conv_class"]["frac"] == conv["frac"]
    if conv.get("traffic"):                       # the committed PMC passes know this instantiation
        assert conv["traffic_over_algorithmic"] == round(conv["traffic"] / conv["algorithmic_bytes_per_launch"], 2) > 1.0
    assert hbm["bound"] == "hbm" and abs(hbm["frac"] - 0.5) < 1e-6


def test_the_whole_n_rank_flow_of_bench_main_runs_on_cpu_with_a_stand_in_engine():
    """The N > 1 path of bench.py has never run on hardware (no multi-GPU box in any round).  FC_BENCH_REHEARSAL=1 swaps the engine for a
    stand-in and runs everything else of main() for real -- self-launch, ranks from the environment, shard_range, per-rank inputs, the step
    loop over micro-batches (2 + 2 + 1 here), gather_codes inside the step, barrier + max-over-ranks timing, ONE JSON line from rank 0 with
    config.ranks_seen / scaling_base -- over gloo on CPU, and checks the gathered codes of all 10 utterances."""
    env = {"FC_BENCH_REHEARSAL": "1", "FC_BENCH_UTTS": "5", "FC_BENCH_SAMPLES": "6400", "FC_BENCH_MICRO": "2"}
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], env_extra=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["ranks_seen"] == 2 and out["config"]["global_utterances"] == 10 and out["config"]["micro_batch"] == 2
    assert "config_c_shard_b128" in out["config"]["scaling_base"]
    assert out["gather_ok"] is True and "REHEARSAL" in out["metric"] and out["value"] == 0.0
    # the self-diagnosing fields of an N > 1 line: per-rank step times, the gather timed alone, the collective library
    mg = out["multi_gpu"]
    assert len(mg["rank_ms_per_step"]) == 2 and mg["rank_ms_per_step_min"] <= mg["rank_ms_per_step_max"]
    assert abs(mg["rank_ms_per_step_max"] - max(mg["rank_ms_per_step"])) < 1e-9 and mg["rank_ms_per_step_max"] <= out["ms_per_step"] * 1.5 + 1.0
    assert mg["gather_ms"] > 0 and mg["gather_bytes_per_rank"] == 5 * 32 * 10 * 8 and mg["backend"] == "gloo" and "rccl_version" in mg
    # and the single-rank flow (no launcher, no gather)
    r1 = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--no-secondary", "--no-cpu-baseline"], env_extra=env)
    assert r1.returncode == 0, r1.stderr[-3000:]
    o1 = _json_lines(r1.stdout)[0]
    assert o1["n_gpus"] == 1 and o1["gather_ok"] is True and o1["config"]["ranks_seen"] == 1 and "multi_gpu" not in o1
