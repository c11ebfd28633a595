"""GPU (-m gpu): the rank program of bench.py under `torch.distributed.run` on ONE GPU -- RCCL initialises, the barrier
and the MAX all-reduce run on the device, and the rank programs of BASELINE configs[3] (every rank owns its N = 30000
bags) and configs[4] (the 64-bag mix split by sharding.assign_bags, through the executor) run on the real kernels.
No scaling number comes out of this; it is there so that the first 8-GPU run cannot fail on plumbing."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("config,extra", [(1, []), (3, ["--streams", "2"]), (4, ["--streams", "3"])])
def test_bench_under_torchrun_world1(config, extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
           "--config", str(config), "--no-cpu-baseline", "--no-extras", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["unit"] == "slides/s" and rec["value"] > 0
    assert rec["config"]["baseline_config_index"] == config
    assert rec["config"]["collectives"]["backend"] == "nccl" and rec["config"]["collectives"]["world"] == 1
