"""GPU: bench.py's N > 1 path rehearsed as the driver launches it -- `python -m torch.distributed.run --nproc-per-node 2
bench.py --gpus 2 ...` in FRESH child processes -- with two ranks sharing the one GPU of the test box (`--dist-backend
gloo`: the collectives of the timing protocol run on the CPU; on a real node the backend is nccl = RCCL).  Checks the
JSON contract: one line from rank 0, n_gpus / global_batch / weak scaling, MAX-over-ranks timing consistent with `value`."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _two_rank_cmd(form, workload, extra):
    """The two launch forms the contract allows: the ranks started by torchrun, or a bare `python bench.py --gpus 2` that
    starts them itself (WORLD_SIZE unset: the driver's own command shape)."""
    tail = ["--gpus", "2", "--dist-backend", "gloo", "--workload", workload, "--no-cpu-baseline", "--no-kernel-events"] + extra
    if form == "self":
        return [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "bench.py")] + tail


def _clean_env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("form,workload", [("torchrun", "guided"), ("self", "guided"), ("torchrun", "sd"), ("self", "adm256")])
def test_bench_two_ranks_prints_one_consistent_json_line(form, workload):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    batch = 8 if workload == "guided" else 2
    cmd = _two_rank_cmd(form, workload, ["--steps", "2", "--warmup", "1", "--batch", str(batch)])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    if form == "self":
        assert "launching 2 ranks" in r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 2 * batch and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["dtype"] == "bf16" and "synthetic" in out["data"] and out["value"] > 0
    assert "secondary" not in out                  # only the default workload (auto) brings the secondary lines along
    # value = units all ranks processed / the (max-over-ranks) time of the K timed steps
    assert abs(out["value"] - 2 * batch / (out["ms_per_step"] * 1e-3)) <= 0.02 * out["value"]
    # the line describes the ranks it timed: who ran where, each rank's own elapsed time, and one all-reduce over the group
    ranks = out["ranks"]
    assert [r_["rank"] for r_ in ranks] == [0, 1] and len({r_["pid"] for r_ in ranks}) == 2
    assert all(r_["device"] and r_["elapsed_s"] > 0 for r_ in ranks)
    assert max(r_["elapsed_s"] for r_ in ranks) <= out["ms_per_step"] * 1e-3 * out["steps"] * 1.001 + 1e-3   # MAX over ranks (+ barrier)
    assert out["collective_check"] == {"backend": "gloo", "allreduce_sum_of_ones": 2.0, "world_size": 2}
    aff = [r_["cpu_affinity"] for r_ in ranks]
    if all(a is not None for a in aff):   # disjoint CPU slices, one per rank
        assert aff[0]["cpus"] != aff[1]["cpus"] and all(a["threads"] >= 1 for a in aff)
    assert out["output_check"]["finite"] is True and out["parity"]["test"].startswith("tests/test_hip_fullsize.py::")


def test_bench_population_two_ranks_lpt_split_and_one_fid_gather_per_epoch():
    """BASELINE configs[2]'s split rehearsed on two ranks (gloo on the one GPU): one EA epoch's population is drawn identically on
    both ranks, LPT-assigned, every candidate evaluated whole on its rank, and ONE all_gather returns the epoch's FIDs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = _two_rank_cmd("self", "population", ["--model", "adm64", "--population", "5", "--images", "8", "--batch", "4",
                                               "--sampler-steps", "3", "--steps", "1", "--warmup", "1"])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["unit"] == "candidates/hour" and out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 1
    cfg = out["config"]
    assert cfg["population"] == 5 and sorted(cfg["candidates_per_rank"]) == [2, 3] and sum(cfg["cost_per_rank"]) == 5 * 3
    assert out["epoch_collective"] == {"op": "all_gather", "backend": "gloo", "world_size": 2, "bytes_per_rank": 5 * 8, "device": "cpu"}
    assert abs(out["value"] - 5 * 3600.0 / (out["ms_per_step"] * 1e-3)) <= 0.02 * out["value"]
    ranks = out["ranks"]
    assert sorted(r_["assigned"] for r_ in ranks) == [2, 3] and all(r_["evaluate_s"] > 0 for r_ in ranks)
    assert all(isinstance(f, float) and f == f for f in out["fid_values_last_epoch"])


def test_bench_single_rank_line_with_the_fid_stage_and_roofline():
    """The N = 1 line as the driver reads it, at a small batch: roofline object from live HIP events, and --with-fid puts the
    Inception + Gram stage inside the step (config.fid_stage_in_step)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline",
           "--with-fid", "--secondary", "adm256"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="4"), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["fid_stage_in_step"] is True and out["unit"] == "images/sec"
    assert "guided" in out["config"]["workload"] and "UNGUIDED" not in out["config"]["workload"]   # auto = the guided headline, or an error
    roof = out["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 2500.0
    assert 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["launches"] > 0
    assert "traffic_source" in roof and (roof["traffic"] is None) == (roof["traffic_source"] is None)
    ev = roof["events_in_timed_region"]
    assert ev["pairs_per_step"] == roof["launches"] // 2 and ev["ms_per_step_without_events"] > 0 and "cost_pct" in ev
    assert out["parity"]["torso"] == "bf16" and "95.5 %" in out["parity"]["last_measured"] and out["hbm_peak_gb"] > 0
    # the secondary line: another BASELINE workload timed in the same process (here the 256x256 one), top-level keys untouched
    sec = out["secondary"]["adm256"]
    assert "error" not in sec, sec
    assert sec["unit"] == "images/sec" and sec["value"] > 0 and sec["steps"] == 2 and "LSUN-256" in sec["workload"]
    assert abs(sec["value"] - 64 / (sec["ms_per_step"] * 1e-3)) <= 0.02 * sec["value"]
    assert 0 < sec["roofline"]["frac"] < 1 and sec["roofline"]["launches"] > 0 and sec["output_check"]["finite"] is True
    assert sec["output_check"]["u8_shape"] == [64, 256, 256, 3]
    chk = out["output_check"]
    assert chk["finite"] is True and chk["u8_shape"] == [8, 64, 64, 3] and chk["u8_std"] > 0


def test_bench_candidate_workload_line():
    """--workload candidate: one whole get_cand_fid per step (here 24 images in batches of 8) -> candidates/hour with the
    reference's reset / sample / fid_time split, finite FIDs, the output check and the roofline of one eager batch."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "candidate", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--images", "24", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="4"), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["unit"] == "candidates/hour" and out["n_gpus"] == 1 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["config"]["images_per_candidate"] == 24 and out["config"]["launch"] == "hipGraph replay"
    ts = out["time_split_s"]
    assert ts["sample_time"] > 0 and ts["fid_time"] > 0 and ts["reset_time"] >= 0
    assert abs(ts["per_candidate"] - out["ms_per_step"] * 1e-3) < 1e-2
    assert ts["reset_time"] + ts["sample_time"] + ts["fid_time"] <= ts["per_candidate"] * 1.05 + 0.05
    assert abs(out["value"] - 3600.0 / ts["per_candidate"]) <= 0.02 * out["value"]
    assert abs(out["images_per_sec"] - 24 / ts["per_candidate"]) <= 0.02 * out["images_per_sec"] + 0.1
    assert len(out["fid_values"]) == 2 and all(f == f for f in out["fid_values"])
    assert out["output_check"]["finite"] is True and out["roofline"]["launches"] > 0


def test_bench_merged_batches_line():
    """--merge-batches K on an image workload: K reference batches per pass (bitwise the same images), value counts all of them."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "unguided", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--merge-batches", "3", "--no-cpu-baseline", "--no-kernel-events"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="4"), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["config"]["batches_per_pass"] == 3 and out["config"]["global_batch"] == 12
    assert abs(out["value"] - 12 / (out["ms_per_step"] * 1e-3)) <= 0.02 * out["value"]
    assert out["output_check"]["finite"] is True and out["output_check"]["u8_shape"] == [4, 64, 64, 3]


def test_more_nccl_ranks_than_gpus_is_refused_with_a_reason():
    """`--gpus 2` with the nccl backend on a box with ONE GPU: RCCL takes one rank per device, so every rank exits at once with a
    message that names the cause (not a hang inside communicator set-up); gloo is the rehearsal backend."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has a GPU per rank")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode != 0
    assert "RCCL takes one rank per device" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
