"""RCCL path on hardware: bench.py's N > 1 code path under torch.distributed.run with the nccl (= RCCL) backend.
world_size 1 runs on the single-GPU test box (init_process_group("nccl"), all_gather_into_tensor, barrier, all_reduce on
device tensors all execute); the world_size-2 test runs where two GPUs are visible and is skipped otherwise."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun_bench(nproc, extra=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1",
           "--batch", "4", "--res", "256", "--no-cpu-baseline", "--no-roofline", "--force-dist", *extra]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_distributed_path_world1_nccl():
    d = _torchrun_bench(1)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["global_batch"] == 4
    assert d["config"]["collective"] == "all_gather_into_tensor over nccl (RCCL), world 1"
    assert d["config"]["gather_side"] == "after"
    # FastVLM-7B width: the gather sits BEFORE the projector (3072-wide tower tokens cross the wire, every rank projects the gathered
    # batch through fvhd_project) - the RCCL leg of that side, at world 1
    d7 = _torchrun_bench(1, ("--hidden", "3584"))
    assert d7["config"]["gather_side"] == "before" and d7["value"] > 0


def test_ttft_distributed_harness_world1_nccl():
    """BASELINE.json configs[3] as a runnable harness (`bench.py --ttft --gpus N`): encode sharded over the ranks -> RCCL all-gather of the
    visual tokens at the projector boundary (before it at the 7B width) -> Qwen2 prefill of the rank's own sequences -> first token.
    World 1 on the single-GPU box: process group, collective, barrier and max-reduce all execute; 2 decoder layers of each width."""
    for hidden, side in ((3584, "before"), (896, "after")):
        d = _torchrun_bench(1, ("--ttft", "--hidden", str(hidden), "--llm-layers", "2", "--batch", "2"))
        c = d["config"]
        assert d["unit"] == "ms" and d["value"] > 0 and d["higher_is_better"] is False and d["n_gpus"] == 1
        assert c["gather_side"] == side and c["world"] == 1 and c["global_batch"] == 2 and c["kv_cache_written"] is True
        assert c["prompt_tokens"] == 24 + 16 and c["prefill_roofline"]["achieved"] > 0
        assert f"configs[3]" in c["workload"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs")
def test_bench_distributed_path_world2_nccl():
    d = _torchrun_bench(2)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0
    d7 = _torchrun_bench(2, ("--hidden", "3584"))             # FastVLM-7B width: gather BEFORE the projector
    assert d7["config"]["gather_side"] == "before"
