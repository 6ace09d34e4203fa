"""The driver's multi-GPU launch line, executed: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py --gpus N --steps K --warmup W` with N = 2 ranks that share the ONE MI355X of a test box over the gloo transport
(CHAM_DIST_BACKEND=gloo: RCCL refuses two ranks per device; everything else - rank / world from the environment, row shards of the global
batch, the gradient exchange hooks, barriers around the timed region, max-over-ranks timing, the exit path - is the code the 8-GPU run
executes).  Round 3 shipped a final barrier inside `if rank == 0` that no test could see because nothing ever ran N > 1; this keeps the
contract under test: exit code 0, exactly ONE JSON line on stdout, the fields the driver reads, weak and strong scaling, the dense all-reduce
and the C1 + C2 exchange (`sparse_rs`), fp32 and the bf16 configuration with its bf16 gradient exchange (BASELINE configs[2]).
(The reference is single-worker by design, README.md:252: there is no reference behaviour for this path, only the contract.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(n, extra, env_extra):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, CHAM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "2",
           "--no-cpu-baseline", "--no-boundary-leg", "--no-ragged-leg", "--no-native-arm", "--no-arms", "--no-pmc"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "rc %d\n%s" % (r.returncode, r.stderr[-3000:])
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), "stdout must carry exactly one JSON line: %r" % lines[:5]
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling,mode,dtype", [("weak", "allreduce", "f32"), ("strong", "allreduce", "f32"), ("weak", "sparse_rs", "f32"),
                                                ("strong", "sparse_rs", "bf16")])
def test_two_ranks_through_the_drivers_launch_line(gpu, scaling, mode, dtype):
    d = _launch(2, ["--scaling", scaling, "--dtype", dtype], dict(CHAM_DP_MODE=mode))
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == scaling and d["unit"] == "sessions/s"
    assert d["metric"] == "NAR training sessions/sec" and d["higher_is_better"] is True and d["data"] == "synthetic"
    per, glob = d["config"]["sessions_per_gpu_per_step"], d["config"]["global_batch"]
    assert (per, glob) == ((256, 512) if scaling == "weak" else (128, 256))
    assert d["config"]["parallelism"] == "dp2" and d["dtype"] == dtype
    # whole-job aggregate over both ranks: global batch x steps / max-over-ranks time
    assert abs(d["value"] - glob / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"] and d["value"] > 100.0
    loss = d["config"]["final_loss"]
    assert 0.5 < loss[1] < 8.0 and abs(loss[0] - loss[1] - loss[2]) < 1e-3, loss       # a finite, plausible loss assembled from both ranks' shares
    assert d["config"]["rnn_coop_spin_timeouts"] == 0
    assert "roofline" in d and d["roofline"]["frac"] > 0.0
    # the N > 1 evidence: what the process group was and what the exchange moved / cost
    dp = d["dp"]
    assert dp["backend"] == "gloo" and dp["world"] == 2 and len(dp["devices"]) == 2 and dp["mode"] == mode
    assert dp["distinct_devices"] == 1                              # (this test: two ranks share the box's one GPU; the driver's run must show N)
    assert dp["grad_dtype"] == ("bf16" if dtype == "bf16" else "f32")
    assert dp["exchange_bytes_per_step"] > 1_000_000 and dp["exchange_ms"] > 0.0 and dp["exchange_steps_timed"] == 5
    assert dp["ring_bus_bytes_per_step"] == dp["exchange_bytes_per_step"]          # 2 (N - 1) / N = 1 at N = 2
    calls = dp["collective_calls_total"]
    if mode == "sparse_rs":        # the C2 pair really ran, once per step (2 warm-up + 3 timed + roofline / host legs + 5 exchange-timing steps)
        assert calls["reduce_scatter"] >= 10 and calls["reduce_scatter"] == calls["all_gather"], calls
    else:
        assert calls["reduce_scatter"] == 0 and calls["all_reduce"] >= 10, calls
