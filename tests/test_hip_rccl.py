"""GPU: every collective of this code base through RCCL, on the one GPU the test box has.

A world-size-1 `nccl` process group (bench.py --force-dist / ADM_FORCE_COLLECTIVES=1) drops the single-rank short-cuts, so
`init_process_group("nccl", device_id=...)`, the barrier / MAX all-reduce of the timing protocol, gather_ranks' all-reduce of ones,
the 32 MiB float64 all_gather of the pooled FID statistics (fid.ActivationAccumulator.pooled) and the per-epoch all_gather of a
population's FIDs (search.EvolutionSearcher.flush_pending) all execute on RCCL with device buffers -- the calls an 8-GPU node
runs, minus the xGMI hops.  Each case is a fresh child process (one process group per process)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", **kw)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ADM_DIST_BACKEND"):
        env.pop(k, None)
    return env


def _bench(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--dist-backend", "nccl", "--no-cpu-baseline",
           "--no-kernel-events"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_timing_protocol_and_pooled_fid_all_gather_on_rccl():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = _bench(["--workload", "adm256", "--with-fid", "--batch", "2", "--steps", "1", "--warmup", "1"])
    assert out["collective_check"] == {"backend": "nccl", "allreduce_sum_of_ones": 1.0, "world_size": 1}
    assert out["ranks"][0]["rank"] == 0 and out["ranks"][0]["device"]
    coll = out["fid_pooled"]["collective"]
    assert coll["op"] == "all_gather" and coll["backend"] == "nccl" and coll["world_size"] == 1
    assert coll["bytes_per_rank"] == 8 * (1 + 2048 + 2048 * 2048) and coll["device"].startswith("cuda")   # the 32 MiB f64 buffer, on the device
    assert out["fid_pooled"]["finite"] is True and out["fid_pooled"]["images_pooled"] == 4
    assert out["output_check"]["finite"] is True


def test_population_epoch_fid_all_gather_on_rccl():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = _bench(["--workload", "population", "--model", "adm64", "--population", "3", "--images", "8", "--batch", "4",
                  "--sampler-steps", "3", "--steps", "1", "--warmup", "1"])
    assert out["collective_check"]["backend"] == "nccl" and out["scaling"] == "strong"
    assert out["epoch_collective"] == {"op": "all_gather", "backend": "nccl", "world_size": 1, "bytes_per_rank": 24,
                                       "device": out["epoch_collective"]["device"]}
    assert out["epoch_collective"]["device"].startswith("cuda")
    assert out["config"]["candidates_per_rank"] == [3] and out["ranks"][0]["assigned"] == 3


FLAGS = ("--image_size 32 --num_channels 32 --num_res_blocks 1 --channel_mult 1,2,2 --attention_resolutions 16,8 "
         "--num_head_channels 32 --class_cond True --learn_sigma True --resblock_updown True --noise_schedule cosine "
         "--use_scale_shift_norm True --use_fp16 True --use_ddim True --classifier_width 64 --classifier_depth 1 "
         "--batch_size 4 --num_samples 8 --time_step 3 --max_epochs 1 --population_num 3 --select_num 2 "
         "--mutation_num 1 --crossover_num 1 --m_prob 0.25 --use_ddim_init_x True --seed 3 --without_classifier True "
         "--features tests.feat_stub:factory").split()


def _search(tmp_path, tag, extra, **env):
    save = str(tmp_path / tag)
    ref = str(tmp_path / "ref.npz")
    if not os.path.exists(ref):
        np.savez(ref, mu=np.zeros(24), sigma=np.eye(24))
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "search_ea.py")] + FLAGS + ["--save_dir", save, "--ref_path", ref] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=_env(PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), MASTER_PORT="29631", **env))
    assert r.returncode == 0, r.stderr[-3000:]
    log = open(os.path.join(save, "log.txt")).read()
    top = re.findall(r"^No\.(\d+) (\[.*?\]) fid = ([-0-9.e+]+)$", log, flags=re.M)
    assert top, log[-2000:]
    return log, [(c, float(f)) for _, c, f in top]


def test_search_cli_pools_and_gathers_through_rccl(tmp_path):
    """scripts/search_ea.py as one nccl rank (dist_util.setup_dist picks nccl on a GPU box): with the collectives forced, the image-
    sharded path pools every candidate's statistics by an RCCL all_gather and the population-parallel path gathers the epoch's FIDs
    by one -- and both print the FIDs of the plain single-process run."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    log0, top0 = _search(tmp_path, "plain", [])
    log1, top1 = _search(tmp_path, "img", [], ADM_FORCE_COLLECTIVES="1")
    log2, top2 = _search(tmp_path, "pop", ["--population_parallel", "True"], ADM_FORCE_COLLECTIVES="1")
    assert "collective:" not in log0
    assert re.search(r"collective: all_gather of the pooled FID statistics, backend nccl, 1 rank\(s\), \d+ B per rank, on cuda", log1), log1[-1500:]
    assert re.search(r"collective: all_gather of \d+ candidate FIDs, backend nccl, 1 rank\(s\), on cuda", log2), log2[-1500:]
    assert [c for c, _ in top0] == [c for c, _ in top1] == [c for c, _ in top2]
    np.testing.assert_array_equal([f for _, f in top1], [f for _, f in top0])
    np.testing.assert_array_equal([f for _, f in top2], [f for _, f in top0])
