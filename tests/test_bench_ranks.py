"""bench.py's N > 1 control flow (rank-local inputs and Philox streams, barriers, the final gather of the samples, the MAX of the
timed region over the ranks, the rank-0 line) rehearsed with two ranks on ONE GPU: every rank on device 0, exchanges host-staged
over gloo (GPT_BENCH_BACKEND=gloo GPT_BENCH_ONE_GPU=1).  The driver runs the same file with one rank per GPU on RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("mode,launcher", [("sample", "torchrun"), ("train", "torchrun"), ("sample", "bare")])
def test_two_rank_bench_line(mode, launcher):
    env = dict(os.environ, GPT_BENCH_BACKEND="gloo", GPT_BENCH_ONE_GPU="1", GPT_BENCH_WATCHDOG="200", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    # "bare": `python bench.py --gpus 2` with no launcher in front starts its own ranks (the reference's main.py:344-354 does)
    front = ([sys.executable] if launcher == "bare" else
             [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29533" if mode == "sample" else "29534"])
    cmd = front + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                   "--warmup", "1", "--no-train", "--no-extra", "--no-roofline", "--no-cpu-baseline", "--repeats", "1"]
    if mode == "train":
        cmd += ["--mode", "train", "--batch", "32"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                    # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    if mode == "sample":
        assert d["config"]["batch_per_gpu"] == 320 and d["scaling"] == "weak"
        # value = the vectors of BOTH ranks over the slowest rank's time
        assert abs(d["value"] - 2 * 320 / (d["ms_per_step"] * 1e-3 * 1000)) < 1e-6 * d["value"]
    else:
        assert d["config"]["global_batch"] == 64 and d["config"]["batch_per_gpu"] == 32


def test_bare_multi_gpu_invocation_spawns_ranks_before_any_gpu_call(monkeypatch):
    """CPU: `python bench.py --gpus 2` without WORLD_SIZE must hand over to torch.distributed.run in a child process (and
    return its exit code) without touching the device in the parent."""
    import importlib
    import subprocess as sp
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None, cwd=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    import torch
    monkeypatch.setattr(sp, "call", fake_call)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *_: pytest.fail("the parent touched the GPU"))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]


def test_headline_survives_sigterm_during_a_secondary_leg():
    """bench.py install_term_guard: rank 0 blocked inside a C call (as in a GPU wait whose peers are gone) is ended by the
    launcher's SIGTERM; the line measured so far must still be printed, with an error object under the running leg, and the exit
    code must be non-zero.  No GPU: the guard and `guarded` are driven directly."""
    import json
    import signal
    import subprocess as sp
    import sys
    import time
    code = r'''
import sys, threading
sys.argv = ["bench.py"]
sys.path.insert(0, %r)
import bench
bench.install_term_guard(0)
line = {"metric": "sampled param-vectors/sec (DDPM)", "value": 55.0}
print("ready", flush=True)
bench.guarded(lambda: threading.Event().wait(60), 0, line, "train", 100.0)     # a lock wait: Python-level handlers cannot run
print("not reached", flush=True)
''' % ROOT
    proc = sp.Popen([sys.executable, "-c", code], stdout=sp.PIPE, stderr=sp.PIPE, text=True, cwd=ROOT)
    assert proc.stdout.readline().strip() == "ready"
    time.sleep(0.5)
    proc.send_signal(signal.SIGTERM)
    out, err = proc.communicate(timeout=30)
    assert proc.returncode == 143, (proc.returncode, out, err[-2000:])
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "not reached" not in out, out
    got = json.loads(lines[0])
    assert got["value"] == 55.0 and "SIGTERM" in got["train"]["error"]
