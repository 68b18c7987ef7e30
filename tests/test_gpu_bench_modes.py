"""The N>1 modes of bench.py, rehearsed with two gloo ranks sharing the one GPU of the test box
(the driver launches the same command with RCCL on 2/4/8 GPUs): rendezvous, the rank-0 broadcast
of the clock-ramp step count (a step of the equalize mode holds a collective, so every rank
must run the same number of steps), barriers, max-over-ranks timing, one JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("config,size", [("c2", 1024), ("c4", 256), ("c5", 1536), ("equalize", 2048)])
def test_bench_two_ranks(config, size):
    env = dict(os.environ, MAGICKHIP_BENCH_RAMP="0.05", MAGICKHIP_BENCH_WATCHDOG="150",
               MAGICKHIP_BENCH_SECONDARY_SIZE="1280")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", config, "--size", str(size),
           "--steps", "2", "--warmup", "1", "--backend", "gloo", "--no-cpu-baseline", "--no-extra"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-800:], out.stderr[-1500:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0
    assert line["scaling"] == ("weak" if config == "c2" else "strong")
    assert line["unit"] == "Mpixels/s" and line["config"]["config"] == config
    assert len(line["per_rank_ms_per_step"]) == 2
    if config == "c2":
        # the default invocation (what the driver launches on 2/4/8 GPUs) also reports the configurations
        # with a collective / with row bands, a few steps each
        multi = line["multi_gpu"]
        assert set(multi) == {"equalize", "c5", "c4"}, multi
        for name, entry in multi.items():
            assert entry["Mpixels_per_s"] > 0 and len(entry["per_rank_ms_per_step"]) == 2, (name, entry)
        assert "used_rccl" in line["collective"] and line["collective"]["equalize_ms_per_step"] > 0
        assert line["collective"]["backend"] == "gloo" and line["collective"]["used_rccl"] is False
