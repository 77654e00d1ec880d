"""CPU: the host-side pieces of bench.py that do not need a GPU -- the layer-skip candidate generator, the output check
that makes a non-finite batch fatal, the per-rank CPU pinning policy and the traffic-source bookkeeping."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_auto_skip_layers_is_deterministic_and_in_range():
    a = bench.auto_skip_layers(58, 5)
    assert a == bench.auto_skip_layers(58, 5) and len(a) == 5
    assert all(len(s) == 6 and s == sorted(set(s)) and 0 <= s[0] and s[-1] < 58 for s in a)
    assert len({tuple(s) for s in a}) > 1          # the steps do not all skip the same layers


def test_check_output_passes_images_and_rejects_nan_or_constant_batches(capsys):
    g = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (4, 8, 8, 3), generator=g, dtype=torch.uint8)
    ok = bench.check_output(torch.randn(4, 3, 8, 8, generator=g), u8)
    assert ok["finite"] is True and ok["u8_shape"] == [4, 8, 8, 3] and ok["u8_checksum"] == int(u8.long().sum()) and ok["u8_std"] > 0
    bad = torch.randn(4, 3, 8, 8, generator=g)
    bad[1, 2, 3, 4] = float("nan")
    with pytest.raises(SystemExit) as e:
        bench.check_output(bad, u8)
    assert e.value.code == 3 and "non-finite" in capsys.readouterr().out
    with pytest.raises(SystemExit):
        bench.check_output(torch.zeros(4, 3, 8, 8), torch.full((4, 8, 8, 3), 127, dtype=torch.uint8))


def test_pin_rank_gives_each_local_rank_its_own_cpu_slice():
    assert bench.pin_rank(0, 1) is None      # one rank: the process is left alone
    if not hasattr(os, "sched_getaffinity") or len(os.sched_getaffinity(0)) < 2:
        pytest.skip("needs >= 2 CPUs and sched_setaffinity")
    code = ("import json, os, sys; sys.path.insert(0, %r); import bench; "
            "print(json.dumps([bench.pin_rank(int(sys.argv[1]), 2), sorted(os.sched_getaffinity(0)), os.environ.get('OMP_NUM_THREADS')]))" % ROOT)
    got = []
    for lr in (0, 1):
        env = {k: v for k, v in os.environ.items() if k != "OMP_NUM_THREADS"}
        r = subprocess.run([sys.executable, "-c", code, str(lr)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append(json.loads(r.stdout.strip().splitlines()[-1]))
    (p0, c0, t0), (p1, c1, t1) = got
    assert p0["n_cpus"] == len(c0) and p1["n_cpus"] == len(c1) and not set(c0) & set(c1)
    assert int(t0) == p0["threads"] >= 1 and int(t1) == p1["threads"] >= 1
    env = dict(os.environ, ADM_BENCH_AFFINITY="0")
    r = subprocess.run([sys.executable, "-c", code, "0"], capture_output=True, text=True, timeout=300, env=env)
    assert json.loads(r.stdout.strip().splitlines()[-1])[0] is None


def test_pmc_traffic_names_its_source():
    v, src = bench.pmc_traffic("guided")
    assert (v is None) == (src is None)
    if v is not None:
        assert v > 0 and "committed constant" in src and "profiles/r0" in src
    assert bench.pmc_traffic("no-such-workload") == (None, None)


def test_rank_cpu_slice_follows_the_gpu_numa_node():
    """8 ranks on a two-socket node (CPUs 0-63 + 128-191 on socket 0, 64-127 + 192-255 on socket 1; GPUs 0-3 on socket 0, 4-7 on
    socket 1): every rank gets CPUs of its own GPU's socket, disjoint from the other ranks'; without topology: contiguous slices."""
    cpus = list(range(256))
    s0 = list(range(0, 64)) + list(range(128, 192))
    s1 = list(range(64, 128)) + list(range(192, 256))
    local_of = lambda i: s0 if i < 4 else s1   # noqa: E731
    got = [bench.rank_cpu_slice(cpus, r, 8, local_of, 8) for r in range(8)]
    assert all(len(g) == 32 for g in got) and len(set(c for g in got for c in g)) == 256
    assert all(set(got[r]) <= set(s0 if r < 4 else s1) for r in range(8))
    flat = [bench.rank_cpu_slice(cpus, r, 8, None, 8) for r in range(8)]
    assert flat[3] == list(range(96, 128)) and len(set(c for g in flat for c in g)) == 256
    # two ranks rehearsing on ONE GPU share its node's CPUs, disjointly
    two = [bench.rank_cpu_slice(cpus, r, 2, lambda i: s0, 1) for r in range(2)]
    assert not set(two[0]) & set(two[1]) and set(two[0]) | set(two[1]) == set(s0)
    assert bench._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    # SMT siblings (cpu c and c + 128) stay together: no two ranks share a physical core
    key = lambda c: c % 128   # noqa: E731
    smt = [bench.rank_cpu_slice(cpus, r, 8, local_of, 8, key) for r in range(8)]
    cores = [set(key(c) for c in g) for g in smt]
    assert all(len(g) == 32 and len(k) == 16 for g, k in zip(smt, cores))
    assert all(not (cores[i] & cores[j]) for i in range(8) for j in range(i))
    assert all(set(smt[r]) <= set(s0 if r < 4 else s1) for r in range(8))


def test_bare_gpus_n_launches_itself_before_any_gpu_call(monkeypatch):
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the driver's own command shape) becomes the launcher: the ranks start as a
    torchrun CHILD process with a clean rendezvous environment, and bench.py exits with the child's code.  No HIP call precedes it
    (this test runs without a GPU: main() must reach self_launch without touching torch.cuda)."""
    import subprocess
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("MASTER_PORT", "1")          # a stale rendezvous variable must not leak into the children
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "2", "--dist-backend", "gloo"])
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: (_ for _ in ()).throw(AssertionError("GPU touched before the launch")))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert cmd[-6:] == ["--gpus", "2", "--steps", "2", "--dist-backend", "gloo"]
    assert "MASTER_PORT" not in seen["env"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_world_size_mismatch_is_an_error(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)


def test_merge_policy_scales_with_the_image_area_and_graph_auto_looks_at_the_merged_pass():
    from autodiffusion_amd.evaluate import graph_auto, merge_policy, pass_cap
    assert (pass_cap(64), pass_cap(128), pass_cap(256), pass_cap(512)) == (256, 128, 64, 16)
    assert merge_policy(64, 100) == (2, 200) and merge_policy(64, 256) == (1, 256) and merge_policy(64, 300) == (1, 300)
    assert merge_policy(128, 32) == (4, 128) and merge_policy(128, 16) == (8, 128)      # the reference's 128x128 batches: 128 images per pass, not 256
    assert merge_policy(256, 36) == (1, 36) and merge_policy(256, 16) == (4, 64)
    assert merge_policy(128, 32, requested=8) == (8, 256) and merge_policy(128, 32, 0, rounds=2) == (2, 64)
    assert graph_auto(64, 200) and not graph_auto(64, 300)          # batch 100 x 2 replays graphs; 3 x 100 does not
    assert graph_auto(128, 64) and not graph_auto(128, 128) and not graph_auto(256, 36)
    assert graph_auto(256, 16)


def test_secondary_lines_keep_the_numbers_and_drop_the_prose():
    line = {"metric": "m", "value": 1.5, "unit": "images/sec", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 10.0, "scaling": "weak",
            "dtype": "bf16", "data": "long prose", "config": {"workload": "w", "global_batch": 64}, "model_tflops": 3.0,
            "roofline": {"bound": "mfma", "kernel": "k", "achieved": 1.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.0004, "launches": 3,
                         "avg_launch_us": 1.0, "avg_launch_gflop": 2.0, "traffic_source": "prose", "how": "prose"},
            "output_check": {"finite": True, "how": "prose"}, "cpu_baseline": {"value": 1}, "hbm_peak_gb": 2.0}
    c = bench.compact(line)
    assert c["value"] == 1.5 and c["workload"] == "w" and c["roofline"]["frac"] == 0.0004 and c["hbm_peak_gb"] == 2.0
    assert "data" not in c and "cpu_baseline" not in c and "how" not in c["roofline"] and c["output_check"] == {"finite": True}
