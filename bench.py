#!/usr/bin/env python3
"""Throughput benchmark of the encode_images() hot path (BASELINE.json metric: images/sec FastViTHD
encode @1024x1024 bf16).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (FastViTHD tower + mlp2x_gelu projector, H=896 = FastVLM-0.5B)
over one synthetic batch of 32 images per GPU, already resident in HBM (BASELINE.json configs[1]).
With N > 1 every rank encodes its own 32 images (weak scaling, no data-path collective inside the
encoder) and the step ends with the single RCCL all-gather of visual tokens at the projector
boundary.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      - for the dominant kernel class: algorithmic FLOPs (or bytes) per launch divided by
                  the average launch duration measured live with HIP events on the launch stream
                  (fvhd_profile_*), against the dense bf16 MFMA peak / HBM3E peak.
  cpu_baseline  - the CPU oracle (a port of the reference's PyTorch path; the reference itself is
                  not present on the GPU box) timed on this host's cores on a bounded sample.
  kernels       - per-class table from the same profiled pass (ms per step, launches, TF/s or GB/s).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0     # dense bf16, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0         # HBM3E spec, same table


def algorithmic_work(B: int, R: int, hidden: int, fused: bool = True, dw_mix=None):
    """Per kernel class: (FLOPs, HBM bytes) of ONE step at batch B, counted algorithmically
    (2*MAC; each activation tensor once in + once out per launch, bf16; weights once).
    dw_mix(H, C) -> bool: the RepMixerBlocks whose dw3x3 and dw7x7 run as ONE launch (class "dw_mix", round 6).  That class carries
    the algorithmic figures of BOTH convolutions (4 tensor passes, although the launch moves 3): the conv-stage total stays the
    SURVEY 8d figure whichever way the kernels are grouped."""
    from ml_fastvlm_amd import fastvithd_spec as spec
    w = {k: [0.0, 0.0] for k in ("stem", "dw3", "dw7", "dw_down", "gemm_fc1", "gemm_fc2", "gemm_1x1", "gemm_qkv",
                                 "gemm_proj", "layernorm", "attention", "head", "projector", "ffn_fused", "dw_mix")}

    def add(k, flops, bytes_):
        w[k][0] += flops
        w[k][1] += bytes_

    def gemm(k, M, N, K, resid=False):
        add(k, 2.0 * M * N * K, 2.0 * (M * K + M * N * (2 if resid else 1) + N * K))

    H = R // 2
    # convolutional_stem in one launch: neither intermediate ([B, R/2, R/2, 96], [B, R/4, R/4, 96]) reaches HBM (1.13x halo recompute
    # of the dense conv is not counted: algorithmic figures)
    add("stem", 2.0 * B * H * H * 96 * 27 + 2.0 * B * (H // 2) ** 2 * 96 * 9, B * (3 * R * R * 2 + (H // 2) ** 2 * 96 * 2))
    H //= 2
    # stem[2] (1x1 + GELU) runs in the same launch since round 4: its FLOPs and weights count, its input / output tensor does not exist in HBM
    add("stem", 2.0 * B * H * H * 96 * 96, 2.0 * 96 * 96)
    for s, (C, depth) in enumerate(zip(spec.EMBED_DIMS, spec.LAYERS)):
        M = B * H * H
        if spec.HAS_CPE[s]:
            add("dw7", 2.0 * M * C * 49, 4.0 * M * C)
        mix = spec.TOKEN_MIXERS[s] == "repmixer" and dw_mix is not None and dw_mix(H, C)
        for _ in range(depth):
            if mix:
                add("dw_mix", 2.0 * M * C * (9 + 49), 8.0 * M * C)
            elif spec.TOKEN_MIXERS[s] == "repmixer":
                add("dw3", 2.0 * M * C * 9, 4.0 * M * C)
            else:
                add("layernorm", 8.0 * M * C, 4.0 * M * C)
                gemm("gemm_qkv", M, 3 * C, C)
                N = H * H
                add("attention", 4.0 * B * (C // 32) * N * N * 32, 2.0 * (M * 3 * C + M * C))
                gemm("gemm_proj", M, C, C, resid=True)
            if not mix:
                add("dw7", 2.0 * M * C * 49, 4.0 * M * C)
            if fused and C <= 384:     # fc1 + GELU + fc2 + layer-scale + residual in one launch, hidden on chip
                add("ffn_fused", 16.0 * M * C * C, 2.0 * (3 * M * C + 8 * C * C))
            else:
                gemm("gemm_fc1", M, 4 * C, C)
                gemm("gemm_fc2", M, C, 4 * C, resid=True)
        if s < len(spec.LAYERS) - 1:
            C2 = spec.EMBED_DIMS[s + 1]
            add("dw_down", 2.0 * (M // 4) * C2 * 49, 2.0 * (M * C + (M // 4) * C2))
            H //= 2
            gemm("gemm_1x1", B * H * H, C2, C2)
    T = H * H
    add("head", 2.0 * B * T * 3072 * 9 + 4.0 * B * 3072 * 192, 2.0 * B * T * (1536 + 3 * 3072))
    gemm("projector", B * T, hidden, 3072)
    gemm("projector", B * T, hidden, hidden)
    return {k: tuple(v) for k, v in w.items()}


PMC_PARTS = {"ffn_fused": ["ffn_fused_c384", "ffn_fused_c192", "ffn_fused_c96"],
             "dw7": ["dw7_mfma_c64 (C = 192, 384)", "dw7_mfma_c96 (C = 96)", "dw7_s1"], "dw3": ["dw3_s1"], "dw_mix": ["dw_mixer_fused"],
             "attention": ["attention"], "dw_down": ["dw_down"], "stem": ["stem"], "layernorm": ["layernorm"],
             "gemm_qkv": ["gemm_plain (qkv)"]}


class PowerSampler:
    """Package power and shader clock of the device HIP runs on while the timed steps run (sysfs hwmon of its PCI address, 25-ms period):
    the hot kernels sit at the 1.4-kW package cap, where wall time follows JOULES per step, not cycles - so the line carries
    them.  Absent hwmon files (or no permission) give `None`."""

    def __init__(self, device_index: int):
        import ctypes
        import glob
        import threading
        self.pw, self.clk, self.cap, self.stop = [], [], None, False
        self.pfile = self.cfile = None
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, device_index) == 0:
                for hw in glob.glob(f"/sys/bus/pci/devices/{buf.value.decode().lower()}/hwmon/hwmon*"):
                    for f in ("power1_average", "power1_input"):
                        if os.path.exists(os.path.join(hw, f)):
                            self.pfile = os.path.join(hw, f)
                            self.cfile = os.path.join(hw, "freq1_input") if os.path.exists(os.path.join(hw, "freq1_input")) else None
                            capf = os.path.join(hw, "power1_cap")
                            self.cap = int(open(capf).read()) / 1e6 if os.path.exists(capf) else None
                            break
                    if self.pfile:
                        break
        except Exception:
            self.pfile = None
        self.thread = threading.Thread(target=self._run, daemon=True) if self.pfile else None

    def _run(self):
        while not self.stop:
            try:
                self.pw.append(int(open(self.pfile).read()) / 1e6)
                if self.cfile:
                    self.clk.append(int(open(self.cfile).read()) / 1e6)
            except Exception:
                pass
            time.sleep(0.025)

    def start(self):
        if self.thread:
            self.thread.start()

    def finish(self, seconds: float, steps: int):
        self.stop = True
        if self.thread:
            self.thread.join(timeout=1)
        if not self.pw:
            return None
        mean = sum(self.pw) / len(self.pw)
        return {"mean_w": round(mean, 1), "max_w": round(max(self.pw), 1), "cap_w": self.cap, "joules_per_step": round(mean * seconds / steps, 3),
                "sclk_mhz": round(sum(self.clk) / len(self.clk)) if self.clk else None, "samples": len(self.pw),
                "source": "sysfs hwmon power1 of the HIP device, sampled every 25 ms over the timed steps"}


def pmc_summary(res: int, batch: int):
    """The newest committed rocprofv3 PMC summary (`profiles/*_pmc_summary.json`: tools/run_pmc.sh + tools/pmc_summary.py, separate
    `--pmc` passes of this same command) whose workload (`_meta`: image size, batch) is THIS run's - counters of another workload
    say nothing about this one, so anything else yields None."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
        prof = json.load(open(f))
        meta = prof.get("_meta", {"res": 1024, "batch": 32})      # summaries older than round 3 were all taken at B = 32 @1024^2
        if int(meta.get("res", 0)) == res and int(meta.get("batch", 0)) == batch:
            return prof, os.path.relpath(f, ROOT)
        return None, None                                          # only the newest summary counts: an older one is of an older binary
    return None, None


def pmc_traffic(prof, kernel_class: str):
    """HBM bytes per launch of a kernel class: 2 x FETCH_SIZE + WRITE_SIZE (KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM":
    gfx950 tallies 128-B read requests at 64 B - calibrated per access pattern in profiles/r04_fetch_calib.log, the summary's `_meta`
    carries the factor per class), averaged over the dispatches of the class' kernels; (bytes, dispatches) or (None, 0)."""
    parts = PMC_PARTS.get(kernel_class)
    if not prof or not parts:
        return None, 0
    tot, n = 0.0, 0
    for cls in parts:
        c = prof.get(cls)
        if not c:
            continue                                 # a class this build does not launch (e.g. no VALU dw7x7 at this batch size)
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            return None, 0
        d = c["FETCH_SIZE"]["dispatches"]
        rf = prof.get("_meta", {}).get("read_factor", {})                 # FETCH_SIZE -> bytes: 2 (wide loads), 1 for the stem's 64-B gather requests
        tot += d * 1024.0 * (rf.get(cls, rf.get("default", 2.0)) * c["FETCH_SIZE"]["per_dispatch"] + c["WRITE_SIZE"]["per_dispatch"])
        n += d
    return (tot / n, n) if n else (None, 0)


def pmc_mfma_busy(prof, pmc_class: str):
    """MFMA-busy fraction of a PMC class: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x dispatch cycles from GRBM_GUI_ACTIVE / 8 XCDs)."""
    c = (prof or {}).get(pmc_class)
    if not c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        return None
    cyc = c["GRBM_GUI_ACTIVE"]["per_dispatch"] / 8.0
    return round(c["SQ_VALU_MFMA_BUSY_CYCLES"]["per_dispatch"] / (1024.0 * cyc), 4) if cyc > 0 else None


GEMM_CLASSES = {"gemm_fc1", "gemm_fc2", "gemm_1x1", "gemm_qkv", "gemm_proj", "attention", "projector", "ffn_fused"}


def cpu_baseline(res: int, hidden: int, budget_s: float = 25.0):
    """Times the reference's CPU path on B=1 images of the same workload: the reference's OWN `MobileCLIPVisionTower` +
    `build_vision_projector` modules (`kind: "reference"`; oracle/ref_import.py imports them from /root/reference in the build
    container, from the archive staged under oracle/_ref/ on the GPU box) - or, when neither is there, the CPU oracle (`kind:
    "port"`: the same ATen calls).  The host of a GPU box can expose far more logical CPUs than a oneDNN convolution of this size
    scales to (256 threads ran 100x slower than 8 in round 1), so a few thread counts are tried and the best is reported together
    with the number of threads it used."""
    from ml_fastvlm_amd import synth
    from oracle import ref_import
    torch.set_flush_denormal(True)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    sd = synth.synthetic_state_dict(1234)
    pj = synth.synthetic_projector_state_dict(hidden, 1234)
    x = synth.synthetic_images(1, res, seed=0)
    kind, what = "port", "torch CPU oracle of the reference path"
    try:
        if ref_import.reference_available():
            tower = ref_import.build_reference_tower(res)
            tower.vision_tower.model.load_state_dict(sd, strict=True)
            proj = ref_import.build_reference_projector(hidden)
            proj.load_state_dict(pj, strict=True)

            def run():
                with torch.no_grad():
                    return proj(tower(x))                # llava_arch.py:141-144 on the reference's own modules
            run()
            kind, what = "reference", "the reference's MobileCLIPVisionTower + mlp2x_gelu projector modules (unmodified), torch CPU"
    except Exception as e:                                # a broken staging must not cost the benchmark line
        print(f"[bench] reference CPU baseline unavailable ({type(e).__name__}: {e}); timing the oracle port", file=sys.stderr)
        kind = "port"
    if kind == "port":
        from oracle import fastvithd_oracle as O

        def run():
            return O.encode_images(x, sd, pj)
    best = None
    t_start = time.perf_counter()
    for threads in sorted({min(avail, t) for t in (8, 16, 32, 64, avail)}):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        run()                                        # warm-up for this thread count
        warm = time.perf_counter() - t0
        if warm > 8.0:                               # pathological oversubscription: do not burn the budget
            continue
        n, t0 = 0, time.perf_counter()
        while n < 3:
            run()
            n += 1
        rate = n / (time.perf_counter() - t0)
        if best is None or rate > best[0]:
            best = (rate, threads, n)
        if time.perf_counter() - t_start > budget_s:
            break
    if best is None:
        return {"value": None, "unit": "images/sec", "cores": avail, "kind": kind, "sample": "every thread count exceeded 8 s per image"}
    cpu = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), cpu)
    except OSError:
        pass
    return {"value": round(best[0], 4), "unit": "images/sec", "cores": best[1], "kind": kind, "cpu": cpu,
            "sample": f"{best[2]} x (1 image {res}x{res}, fp32, {what}, best of thread counts <= {avail} logical CPUs; {best[1]} threads used)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--hidden", type=int, default=896, help="LLM hidden size of the projector (896 = Qwen2-0.5B)")
    ap.add_argument("--graph", action="store_true", help="hipGraph replay of the tower's interior launches (fvhd_set_graph)")
    ap.add_argument("--attn-fp8", action="store_true", help="e4m3 MFMA operands in the attention cores (BASELINE.json configs[4] as named; opt-in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--tower-only", action="store_true", help="time the vision tower without the projector")
    ap.add_argument("--force-dist", action="store_true", help="take the N > 1 code path (process group, collectives, barriers) even at WORLD_SIZE = 1")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of the N > 1 path: nccl = RCCL over xGMI (the product path); gloo = host-staged all-gather, for running "
                         "several ranks on ONE GPU (tests: RCCL refuses two ranks on one device) - never a performance number")
    ap.add_argument("--same-device", action="store_true", help="with --backend gloo: every rank uses cuda:0 (several ranks on a one-GPU box)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="default run: skip the `c4_1gpu` and `c5` objects (BASELINE configs[3] single-GPU leg, configs[4]) appended to the throughput line")
    ap.add_argument("--ttft-llm", default="kernels", choices=["kernels", "kernels-graph", "hf-graph", "hf-eager"],
                    help="--ttft: prefill on the hand-written kernels (fvhd_llm_prefill; default), the same as one hipGraph, or the stock transformers module (graph / eager)")
    ap.add_argument("--ttft", action="store_true", help="report time-to-first-token of FastVLM prefill instead (tools/ttft.py, BASELINE configs[2]; with --gpus N / "
                                                        "--hidden 3584: configs[3] - encode sharded over the ranks, RCCL all-gather at the projector boundary, data-parallel prefill)")
    ap.add_argument("--no-ttft", action="store_true", help="default run: skip the `ttft` object (BASELINE configs[2], B = 8, Qwen2-0.5B widths) appended to the throughput line")
    ap.add_argument("--llm-layers", type=int, default=0, help="--ttft: truncate the decoder stack to this many layers (tests; 0 = the published depth)")
    args = ap.parse_args()
    args.batch_given = any(a == "--batch" or a.startswith("--batch=") for a in sys.argv[1:])

    t_start = time.perf_counter()

    def trace(msg):                              # FVHD_BENCH_TRACE=1: where the wall time of a run goes (stderr)
        if os.environ.get("FVHD_BENCH_TRACE") == "1":
            print(f"[bench {time.perf_counter() - t_start:8.2f}s] {msg}", file=sys.stderr, flush=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path)")
    if world > 1:
        # N ranks build their synthetic weights (and pack them on the host) at the same time: each keeps to its share of the cores instead of
        # N OpenMP teams of all cores fighting (8 ranks on one box: minutes of start-up otherwise; nothing inside the timed region runs on the CPU)
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))
    if args.same_device:
        if args.backend != "gloo":
            raise SystemExit("--same-device needs --backend gloo (RCCL refuses two ranks on one device)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    multi = world > 1 or (args.force_dist and "RANK" in os.environ)      # the distributed code path (also at world 1 when forced)
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        trace("init_process_group ...")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        trace("process group up")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import ml_fastvlm_amd as fv
    from ml_fastvlm_amd import distributed as D
    from ml_fastvlm_amd import synth

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    if args.ttft:                                # BASELINE.json configs[2] / [3]: a latency metric, its own JSON line (tools/ttft.py)
        import ttft
        Bt = args.batch if args.batch_given else 8
        r = ttft.measure(Bt, args.res, args.hidden, args.steps, args.warmup, dev, args.graph, llm_mode=args.ttft_llm,
                         dist_ctx=(rank, world) if multi else None, llm_layers=args.llm_layers)
        if rank == 0:
            cfgno = 3 if (args.hidden == 3584 or multi) else 2
            print(json.dumps({"metric": f"TTFT FastVLM prefill, batch={Bt}/GPU @{args.res}x{args.res} bf16", "value": r["ttft_ms_median"], "unit": "ms",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ttft_ms_median"], "higher_is_better": False,
                              "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                              "config": {"workload": f"BASELINE.json configs[{cfgno}]: encode_images {'(sharded over the ranks, visual tokens all-gathered ' + r['gather_side'] + ' the projector) ' if multi else ''}"
                                                     "-> embedding splice -> Qwen2 prefill of the rank's own sequences -> first token, "
                                                     "qwen_2 prompt around one <image>, synthetic ids/images, random weights", "parallelism": f"dp{world}", **r}}))
        if multi:
            dist.destroy_process_group()
        return

    B, R, Hd = args.batch, args.res, args.hidden
    tower = fv.MobileCLIPVisionTower(f"mobileclip_l_{R}", SimpleNamespace(unfreeze_mm_vision_tower=False, mm_vision_hip_graph=args.graph,
                                                                          mm_vision_attention_fp8=True if args.attn_fp8 else None))
    tower.vision_tower.model.load_state_dict(synth.synthetic_state_dict(1234), strict=True)
    proj = fv.build_vision_projector(SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=3072, hidden_size=Hd))
    proj.load_state_dict(synth.synthetic_projector_state_dict(Hd, 1234), strict=True)
    tower, proj = tower.to(dev, torch.bfloat16), proj.to(dev, torch.bfloat16)
    trace("tower + projector built")
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    images = torch.rand((B, 3, R, R), generator=g).to(dev, torch.bfloat16)      # synthetic, in [0,1), HBM-resident

    @torch.no_grad()                             # inference (predict.py:55 runs generate() under torch.inference_mode())
    def local_step():
        return tower(images) if args.tower_only else fv.encode_images(tower, proj, images)

    side = D.gather_side(Hd)                     # 0.5B / 1.5B: gather the projected tokens; 7B (H = 3584 > 3072): gather before the projector

    def _all_gather(t):                          # (force: the collective is issued at world 1 too)
        return D.all_gather_tokens(t, B * world, force=True)

    @torch.no_grad()
    def step():
        if not multi:
            return local_step()
        if args.tower_only or side == "after":
            return _all_gather(local_step())
        return fv.project(tower, proj, _all_gather(tower(images)))      # fvhd_project: the library's GEMMs, never torch.nn.Linear

    # explicit range calibration of the half-precision fused ConvFFN (the default "auto" would audit the first batches it sees -
    # two extra eager passes that belong in front of the warm-up, not inside a short timed region); the range guard stays on
    tower.calibrate(images)
    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    trace("warm-up done")
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    power = PowerSampler(local_rank) if rank == 0 else None
    if power:
        power.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    power_obj = power.finish(dt, args.steps) if power else None
    trace("timed region done")
    assert torch.isfinite(out.float()).all()
    if multi:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    result = {
        "metric": f"images/sec FastViTHD encode_images @{R}x{R} bf16",
        "value": round(world * B * args.steps / dt, 2),
        "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[{1 if R == 1024 else 4 if R == 1536 else '-'}]: FastViTHD encoder{'' if args.tower_only else ' + mlp2x_gelu projector (H=%d)' % Hd}, "
                               f"batch={B}/GPU synthetic {R}x{R} bf16 images in [0,1), seeded synthetic weights, "
                               f"{('tokens all-gathered over RCCL ' + side + ' the projector') if multi else 'single GPU'}",
                   "global_batch": B * world, "image_size": R, "tokens_per_image": (R // 64) ** 2, "parallelism": f"dp{world}",
                   "hip_graph": bool(args.graph), "attention_operands": "e4m3" if args.attn_fp8 else "bf16",
                   "ffn_precision": "auto (calibrated on the benchmark batch), range guard on",
                   **({"collective": f"all_gather_into_tensor over {'nccl (RCCL)' if args.backend == 'nccl' else 'gloo (host-staged; test only)'}, world {world}",
                       "gather_side": side} if multi else {})},
    }
    result["power"] = power_obj      # package power / joules per step over the timed region (None without hwmon access)

    if rank == 0 and not args.no_roofline:
        ctx = tower._context()
        ctx.profile_enable(True)
        ctx.profile_reset()
        psteps = max(2, min(5, args.steps))
        for _ in range(psteps):
            local_step()                             # rank-local: no collective in the profiled pass
        torch.cuda.synchronize()
        prof = ctx.profile_read()
        ctx.profile_enable(False)
        from ml_fastvlm_amd import _lib as _fl
        mixed = prof.get("dw_mix", (0, 0))[1] > 0              # the tower's own rule (fvhd_dw3_dw7_supported) when the class was launched at all
        work = algorithmic_work(B, R, Hd, fused=prof.get("ffn_fused", (0, 0))[1] > 0,
                                dw_mix=(lambda H_, C_: bool(_fl.load().fvhd_dw3_dw7_supported(B, H_, H_, C_, 0))) if mixed else None)
        table, total_ms = {}, 0.0
        for k, (ms, n) in prof.items():
            if n == 0:
                continue
            ms_step = ms / psteps
            total_ms += ms_step
            fl, by = work[k]
            table[k] = {"ms_per_step": round(ms_step, 3), "launches_per_step": n // psteps,
                        "tflops": round(fl / ms_step / 1e9, 1), "gbs": round(by / ms_step / 1e6, 1)}
        dom = max(table, key=lambda k: table[k]["ms_per_step"])
        n_dom = table[dom]["launches_per_step"]
        avg_ms = table[dom]["ms_per_step"] / n_dom
        if dom in GEMM_CLASSES:
            ach = work[dom][0] / n_dom / (avg_ms * 1e-3) / 1e12
            result["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS,
                                  "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                                  "avg_launch_ms": round(avg_ms, 4), "flops_per_launch": work[dom][0] / n_dom}
        else:
            ach = work[dom][1] / n_dom / (avg_ms * 1e-3) / 1e9
            result["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                                  "avg_launch_ms": round(avg_ms, 4), "bytes_per_launch": work[dom][1] / n_dom}
        prof_pmc, src = pmc_summary(R, B)
        traffic, _ = pmc_traffic(prof_pmc, dom)
        result["roofline"]["traffic"] = None if traffic is None else round(traffic)
        result["roofline"]["traffic_note"] = (f"no committed PMC summary of this workload (res {R}, batch {B})" if traffic is None else
                                              f"HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC, {src}); algorithmic bytes per launch "
                                              f"{work[dom][1] / n_dom:.4g}")
        # the two north_star targets, readable off this line: conv stages against the HBM peak, attention / projector GEMMs against the MFMA peak
        conv = [k for k in ("stem", "dw3", "dw7", "dw_mix", "dw_down") if k in table]
        conv_ms = sum(table[k]["ms_per_step"] for k in conv)
        conv_by = sum(work[k][1] for k in conv)
        conv_tr = [pmc_traffic(prof_pmc, k) for k in conv]
        conv_cnt = None if any(t is None for t, _ in conv_tr) else sum(t * table[k]["launches_per_step"] for (t, _), k in zip(conv_tr, conv))
        result["conv_stage"] = {"classes": conv, "ms_per_step": round(conv_ms, 3), "algorithmic_bytes_per_step": conv_by,
                                "gbs": round(conv_by / conv_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "frac": round(conv_by / conv_ms / 1e6 / HBM_PEAK_GBS, 4),
                                "counter_bytes_per_step": None if conv_cnt is None else round(conv_cnt),
                                "counter_gbs": None if conv_cnt is None else round(conv_cnt / conv_ms / 1e6, 1), "target_frac": 0.60}
        att = [k for k in ("gemm_qkv", "attention", "gemm_proj", "projector") if k in table]
        att_ms = sum(table[k]["ms_per_step"] for k in att)
        att_fl = sum(work[k][0] for k in att)
        result["attention_block"] = {"classes": {k: {"tflops": table[k]["tflops"], "frac": round(table[k]["tflops"] / MFMA_PEAK_TFLOPS, 4)} for k in att},
                                     "ms_per_step": round(att_ms, 3), "tflops": round(att_fl / att_ms / 1e9, 1), "peak": MFMA_PEAK_TFLOPS,
                                     "frac": round(att_fl / att_ms / 1e9 / MFMA_PEAK_TFLOPS, 4), "target_frac": 0.40,
                                     "mfma_busy_pmc": {k: pmc_mfma_busy(prof_pmc, c) for k, c in (("gemm_qkv", "gemm_plain (qkv)"), ("attention", "attention"),
                                                                                                  ("gemm_proj", "gemm_resid (fc2 / proj)"))}}
        result["kernels"] = table
        result["kernel_ms_per_step_profiled"] = round(total_ms, 3)
        tot_fl = sum(v[0] for v in work.values())
        result["whole_step_tflops"] = round(tot_fl / (dt / args.steps) / 1e12, 1)

    if rank == 0 and world == 1 and not args.no_ttft and R == 1024 and not args.tower_only:
        # the second half of the BASELINE metric on the same line (configs[2]: FastVLM-0.5B prefill TTFT, B = 8): encode_images ->
        # embedding splice -> Qwen2-0.5B prefill on fvhd_llm_prefill (KV cache written) -> first token on the host.  ~3 s of the run.
        try:
            import ttft
            t = ttft.measure(8, R, 896, steps=10, warmup=2, dev=dev, llm_mode="kernels")
            result["ttft"] = {"metric": "TTFT FastVLM-0.5B prefill, batch=8 @1024x1024 bf16", "value": t["ttft_ms_median"], "unit": "ms",
                              "higher_is_better": False, "workload": "BASELINE.json configs[2]", **t}
        except Exception as e:                                # the latency leg must not cost the throughput line
            result["ttft"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        trace("ttft leg done")

    if rank == 0 and world == 1 and not args.no_extra_configs and not args.no_ttft and R == 1024 and not args.tower_only and B == 32 and Hd == 896:
        # the remaining BASELINE.json configs on the driver's clock, a few seconds each:
        # c4_1gpu - the single-GPU leg of configs[3]: 8 images, FastVLM-7B widths (Qwen2-7B, 28 layers, 7.6 B random bf16 parameters), TTFT
        #           with the gather side of the 7B model's projector; the 8-GPU all-gather leg is `bench.py --ttft --hidden 3584 --gpus 8`
        # c5      - configs[4]: 1536 x 1536, batch 16, e4m3 MFMA operands in the attention cores (`--res 1536 --batch 16 --attn-fp8`)
        del out
        try:
            import ttft
            t = ttft.measure(8, R, 3584, steps=5, warmup=2, dev=dev, llm_mode="kernels")
            result["c4_1gpu"] = {"metric": "TTFT FastVLM-7B prefill, batch=8 @1024x1024 bf16, one GPU's share of configs[3]", "value": t["ttft_ms_median"],
                                 "unit": "ms", "higher_is_better": False, "workload": "BASELINE.json configs[3], single-GPU leg (no collective)", **t}
        except Exception as e:
            result["c4_1gpu"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
        trace("c4_1gpu leg done")
        try:
            t5 = fv.MobileCLIPVisionTower("mobileclip_l_1536", SimpleNamespace(unfreeze_mm_vision_tower=False, mm_vision_attention_fp8=True, mm_vision_max_batch=16))
            t5.vision_tower.model.load_state_dict(synth.synthetic_state_dict(1234), strict=True)
            t5 = t5.to(dev, torch.bfloat16)
            x5 = torch.rand((16, 3, 1536, 1536), generator=torch.Generator(device="cpu").manual_seed(2000)).to(dev, torch.bfloat16)
            with torch.no_grad():
                t5.calibrate(x5)
                for _ in range(2):
                    o5 = fv.encode_images(t5, proj, x5)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    o5 = fv.encode_images(t5, proj, x5)
                torch.cuda.synchronize()
                d5 = time.perf_counter() - t0
            assert torch.isfinite(o5.float()).all()
            result["c5"] = {"metric": "images/sec FastViTHD encode_images @1536x1536, e4m3 attention operands", "value": round(16 * 5 / d5, 2), "unit": "images/sec",
                            "higher_is_better": True, "ms_per_step": round(1e3 * d5 / 5, 3), "steps": 5, "warmup": 2, "batch": 16,
                            "workload": "BASELINE.json configs[4]: FastViTHD at 1536x1536, batch=16, fp8 (e4m3) MFMA attention path, 1 GPU",
                            "tokens_per_image": 576, "attention_operands": "e4m3", "dtype": "bf16 (attention operands e4m3)"}
            del t5, x5, o5
        except Exception as e:
            result["c5"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
        trace("c5 leg done")

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(R, Hd)

    if rank == 0:
        print(json.dumps(result))
    if multi:
        trace("destroy_process_group ...")
        dist.destroy_process_group()
    trace("done")


if __name__ == "__main__":
    main()
