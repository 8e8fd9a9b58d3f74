"""`python bench.py --gpus N` must run N ranks by itself (the shape of the driver's command) -- tested both ways:

* without a GPU (here): the launcher starts N children with RANK / WORLD_SIZE set, every child fails loudly ("needs a HIP
  device"), the launcher stops the rest and exits non-zero -- no silent one-rank run, no hang;
* on the GPU box: the 2-rank rehearsal on ONE GPU (gloo for the timing collectives, both ranks on device 0): the N-rank code
  path of bench.py end to end -- interleaved 16-row band shards, gather, de-interleave -- with the gathered frame compared byte
  for byte against the single-device frame before anything is timed.  RCCL refuses two ranks on one device, so this goes
  through the torch.distributed exchange; the C-ABI exchange's R > 1 transfers need R GPUs (the driver's scaling run).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_launcher_starts_ranks_and_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("the no-device behaviour is what is tested here")
    env = dict(os.environ, ARCTIC_BENCH_BACKEND="gloo", ARCTIC_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu", "--scale", "0.05"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0
    assert "needs a HIP device" in out.stderr
    assert "the other ranks were stopped" in out.stderr or "exited with code" in out.stderr
    assert out.stdout.strip() == ""          # no result line from a failed run


@pytest.mark.parametrize("visible", [None, "0"])
def test_launcher_refuses_more_ranks_than_devices(visible):
    """the launcher counts the devices without a HIP call (the KFD topology, or the *_VISIBLE_DEVICES list when one is set)"""
    import torch
    if visible is None and torch.cuda.device_count() >= 2:
        pytest.skip("node has 2+ devices")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "ARCTIC_BENCH_SHARE_GPU", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        env.pop(k, None)
    if visible is not None:
        env["HIP_VISIBLE_DEVICES"] = visible
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0 and "one GPU per rank" in out.stderr and out.stdout.strip() == ""


def test_visible_device_lists_compose(monkeypatch):
    """ROCR_VISIBLE_DEVICES and HIP_VISIBLE_DEVICES compose (HIP indexes into what ROCr shows): the smallest count holds, and never more than
    the topology's GPUs this process may open"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    topo = bench.count_gpus_without_hip()
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1,2")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    got = bench.count_gpus_without_hip()
    assert got is not None and got <= 1 and (topo is None or got <= topo)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.count_gpus_without_hip() == 0


def test_launcher_refuses_to_start_ranks_under_a_profiler():
    """under rocprofv3 the preloaded tool library has initialised the GPU before bench.py starts: no fork + exec from there"""
    env = dict(os.environ, ROCPROFILER_REGISTER_FORCE_LOAD="1")
    for k in ("WORLD_SIZE", "RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0 and "under a profiler" in out.stderr and out.stdout.strip() == ""


@pytest.mark.gpu
def test_two_rank_rehearsal_on_one_gpu(hip):
    env = dict(os.environ, ARCTIC_BENCH_BACKEND="gloo", ARCTIC_BENCH_SHARE_GPU="1", ARCTIC_BENCH_VERIFY="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "strong"
    assert line["config"]["exchange_path"] == "torch.distributed gather + index_copy_"
    assert line["config"]["verified_against_single_device_frame"] is True
    assert "identical to the single-device frame: True" in out.stderr
    assert line["config"]["shaded_pixels"] == 3840 * 2160
    assert line["value"] > 0
