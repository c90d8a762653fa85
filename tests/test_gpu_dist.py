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


# ---- world size 2 with the REAL tower, on ONE GPU ---------------------------------------------------------------------
# RCCL refuses two ranks on one device, so the two processes share cuda:0 under the gloo backend (the visual tokens are staged through the
# host for the collective: distributed.all_gather_into).  What this proves is everything EXCEPT the wire: shard order, ragged shards, the
# gather side, the library projector after the gather - bit for bit against a single-process encode of the whole batch (batch-invariant
# kernel selection: an image gives the same bits in any batch; reference semantics: images are independent, mobileclip_encoder.py:78-83,
# llava_arch.py:154-160).
def _real_tower_worker(rank, world, port, global_batch, q):
    import torch.distributed as dist
    from types import SimpleNamespace
    import ml_fastvlm_amd as fv
    from ml_fastvlm_amd import distributed as D
    from ml_fastvlm_amd import synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))      # `world` processes build their synthetic weights at once: no oversubscription
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        tower = fv.MobileCLIPVisionTower("mobileclip_l_256", SimpleNamespace(unfreeze_mm_vision_tower=False, mm_vision_batch_invariant=True))
        tower.vision_tower.model.load_state_dict(synth.synthetic_state_dict(1234, "mild"), strict=True)
        tower = tower.to(dev, torch.bfloat16)
        images = synth.synthetic_images(global_batch, 256, seed=23).to(dev, torch.bfloat16)       # the same batch on both ranks
        ok, notes = True, []
        with torch.no_grad():
            for hidden, side in (((896, "after"), (3584, "before")) if world <= 2 else (((896, "after"),) if global_batch % 2 == 0 else ((3584, "before"),))):
                proj = fv.build_vision_projector(SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=3072, hidden_size=hidden))
                proj.load_state_dict(synth.synthetic_projector_state_dict(hidden, 1234), strict=True)
                proj = proj.to(dev, torch.bfloat16)
                want = fv.encode_images(tower, proj, images)                                    # single-process result for the whole batch
                local = D.shard_images(images)
                assert D.gather_side(hidden) == side
                got = D.encode_images_tower_sharded(tower, proj, local, global_batch)
                same = got.shape == want.shape and torch.equal(got, want)
                notes.append(f"H={hidden} side={side} shard={tuple(local.shape)} equal={same}")
                ok = ok and same
                # ... and the other side explicitly: the two orders of gather and projector agree bit for bit as well
                other = D.encode_images_tower_sharded(tower, proj, local, global_batch, side="before" if side == "after" else "after")
                ok = ok and torch.equal(other, want)
        torch.cuda.synchronize()
        q.put((rank, bool(ok), notes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [4, 5])      # equal shards, and a ragged split (3 + 2: padded for the collective, trimmed after)
def test_world2_real_tower_on_one_gpu_equals_single_process(global_batch):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_tower_worker, args=(r, 2, port, global_batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    print(res)
    assert [(r, ok) for r, ok, _ in res] == [(0, True), (1, True)], res


@pytest.mark.parametrize("global_batch", [8, 11])     # one image per rank, and a ragged split (2 + 2 + 2 + 1 + 1 + 1 + 1 + 1)
def test_world8_real_tower_on_one_gpu_equals_single_process(global_batch):
    """The world size of the BASELINE metric (8 ranks), all on cuda:0 under gloo: shard bounds, padding and trimming of ragged shards, both
    gather sides and the library projector behind the gather at the REAL world size - everything of the 8-GPU run except the wire
    (VERDICT r5 item 8).  Every rank checks the gathered result against its own single-process encode of the whole batch, bit for bit."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_tower_worker, args=(r, 8, port, global_batch, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    print(res)
    assert [(r, ok) for r, ok, _ in res] == [(r, True) for r in range(8)], res


def test_bench_world8_on_one_gpu_gloo():
    """`bench.py --gpus 8` - the driver's 8-GPU command line - with the eight ranks on cuda:0 under gloo: process group, shard, gather, barrier,
    max-reduce and the JSON line at world 8 (one 256^2 image per rank; never a performance number)."""
    d = _torchrun_bench(8, ("--backend", "gloo", "--same-device", "--batch", "1"))
    c = d["config"]
    assert d["n_gpus"] == 8 and c["global_batch"] == 8 and "gloo" in c["collective"] and "world 8" in c["collective"] and d["value"] > 0
    assert d["scaling"] == "weak"


def test_ttft_harness_world2_on_one_gpu_gloo():
    """`bench.py --ttft --gpus 2` (BASELINE.json configs[3] harness) with two ranks on cuda:0 under gloo: encode sharded over the ranks ->
    all-gather at the projector boundary (before it at the 7B width) -> every rank prefills its own sequences -> max over ranks."""
    for hidden, side in ((3584, "before"), (896, "after")):
        d = _torchrun_bench(2, ("--ttft", "--hidden", str(hidden), "--llm-layers", "2", "--batch", "2", "--backend", "gloo", "--same-device"))
        c = d["config"]
        assert d["n_gpus"] == 2 and d["unit"] == "ms" and d["value"] > 0
        assert c["world"] == 2 and c["global_batch"] == 4 and c["gather_side"] == side and "gloo" in c["collective"]
    d = _torchrun_bench(2, ("--backend", "gloo", "--same-device"))               # the throughput line's N > 1 path, two ranks on one device
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and "gloo" in d["config"]["collective"] and d["value"] > 0
