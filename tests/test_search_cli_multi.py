"""GPU: scripts/search_ea.py end to end as separate processes -- one rank, two ranks sharding every candidate's IMAGES
(one pooled all-gather per candidate) and two ranks sharding whole CANDIDATES (--population_parallel) -- must print the same
"top" report.  Two ranks share the test box's one GPU, so the process group is gloo (ADM_DIST_BACKEND); on a node it is RCCL."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FLAGS = ("--image_size 32 --num_channels 32 --num_res_blocks 1 --channel_mult 1,2,2 --attention_resolutions 16,8 "
         "--num_head_channels 32 --class_cond True --learn_sigma True --resblock_updown True --noise_schedule cosine "
         "--use_scale_shift_norm True --use_fp16 True --use_ddim True --classifier_width 64 --classifier_depth 1 "
         "--image_size 32 --batch_size 4 --num_samples 8 --time_step 3 --max_epochs 2 --population_num 4 --select_num 2 "
         "--mutation_num 1 --crossover_num 1 --m_prob 0.25 --use_ddim_init_x True --seed 3 --without_classifier True "
         "--features tests.feat_stub:factory").split()


def _run(tmp_path, tag, nproc, extra):
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    save = str(tmp_path / tag)
    ref = str(tmp_path / "ref.npz")
    if not os.path.exists(ref):
        np.savez(ref, mu=np.zeros(24), sigma=np.eye(24))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "scripts", "search_ea.py")] + FLAGS + ["--save_dir", save, "--ref_path", ref] + extra
    env = dict(os.environ, ADM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    log = open(os.path.join(save, "log.txt")).read()
    top = re.findall(r"^No\.(\d+) (\[.*?\]) fid = ([-0-9.e+]+)$", log, flags=re.M)
    assert top, log[-2000:]
    last = [t for t in top if True][-min(len(top), 8):]
    return log, [(c, float(f)) for _, c, f in top]


def test_search_cli_one_rank_vs_image_sharded_vs_population_parallel(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    log1, top1 = _run(tmp_path, "one", 1, [])
    log2, top2 = _run(tmp_path, "img", 2, [])
    log3, top3 = _run(tmp_path, "pop", 2, ["--population_parallel", "True"])
    assert "epoch = 1 : top" in log1 and log1.count("sampling complete") >= 4
    assert [c for c, _ in top1] == [c for c, _ in top2] == [c for c, _ in top3]        # same candidates, same ranking
    f1, f2, f3 = (np.array([f for _, f in t]) for t in (top1, top2, top3))
    np.testing.assert_allclose(f2, f1, rtol=1e-7, atol=1e-9)   # same images, pooled sums in a different order
    np.testing.assert_array_equal(f3, f1)                        # whole candidates on one rank: the single-process arithmetic
    # ranks > 0 log to their own file, not to log.txt / stdout
    assert os.path.exists(os.path.join(str(tmp_path / "img"), "log-rank001.txt"))
    assert log2.count("epoch = 0 : top") == 1


def _run_bundled(tmp_path, tag, nproc, opt_in=True):
    ref = str(tmp_path / "ref2048.npz")
    if not os.path.exists(ref):
        rng = np.random.RandomState(0)
        a = rng.randn(2048, 2048) / 45.0
        np.savez(ref, mu=rng.randn(2048) * 0.1, sigma=a @ a.T + 0.1 * np.eye(2048))
    flags = [f for f in FLAGS]
    i = flags.index("--features")
    del flags[i:i + 2]
    save = str(tmp_path / tag)
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "scripts", "search_ea.py")] + flags + [
               "--save_dir", save, "--ref_path", ref, "--fid_on_device", "True", "--max_epochs", "1"] + (
                   ["--inception_random", "True"] if opt_in else [])
    env = dict(os.environ, ADM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    if not opt_in:   # neither --features nor --inception_path: the CLI refuses to rank candidates on random-weight features
        assert r.returncode != 0 and "--inception_random" in r.stderr, r.stderr[-2000:]
        return None
    assert r.returncode == 0, r.stderr[-3000:]
    log = open(os.path.join(save, "log.txt")).read()
    assert "RANDOM weights" in log
    # every FID line of the log carries the tag, next to the value
    top = re.findall(r"^No\.(\d+) (\[.*?\]) fid = ([-0-9.e+]+) \[FID on RANDOM Inception weights: not a quality metric\]$", log, flags=re.M)
    assert top and all(np.isfinite(float(f)) for _, _, f in top), log[-2000:]
    assert len(top) == len(re.findall(r"^No\.\d+ .* fid = ", log, flags=re.M))
    assert all("RANDOM Inception" in ln for ln in log.splitlines() if ln.startswith("cand: ") and ", fid: " in ln)
    return [(c, float(f)) for _, c, f in top]


def test_search_cli_with_the_bundled_inception_extractor(tmp_path):
    """No --features: the HIP Inception-v3 pool3 extractor scores the candidates (random weights here -- the checkpoint
    is not in the image -- so the FID values mean nothing; the path from uint8 batches to the "top" report is what runs),
    on one rank and with every candidate's images sharded over two ranks (features are per image, the float64 sums pooled)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    assert _run_bundled(tmp_path, "inc0", 1, opt_in=False) is None
    top1 = _run_bundled(tmp_path, "inc1", 1)
    top2 = _run_bundled(tmp_path, "inc2", 2)
    assert [c for c, _ in top1] == [c for c, _ in top2]
    np.testing.assert_allclose([f for _, f in top2], [f for _, f in top1], rtol=1e-6, atol=1e-6)
