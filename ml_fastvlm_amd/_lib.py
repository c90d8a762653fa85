"""ctypes binding of libfvhd.so (include/fvhd.h).  This is the whole Python<->native boundary:
plain C pointers and sizes, no torch types.  There is no CPU fallback: if the library is missing
the import fails loudly, and creating a context without a HIP device is an error."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
# FVHD_LIB: another build of the same ABI (the ablation library of `python -m ml_fastvlm_amd.build` with FVHD_FFN_ABLATE=1)
LIB_PATH = os.environ.get("FVHD_LIB") or os.path.join(_HERE, "libfvhd.so")

ABI_VERSION = 500               # FVHD_VERSION of the include/fvhd.h this stub was written against (major = ABI_VERSION // 100)
F32, F16, BF16 = 0, 1, 2
FFN_HALF, FFN_BF16 = 0, 1        # precision of the fused ConvFFN's hidden activation (include/fvhd.h)
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_LS_RESID, EPI_RESID, EPI_SWIGLU = 0, 1, 2, 3, 4, 5

_lib = None


class FvhdError(RuntimeError):
    pass


def _declare(lib) -> None:
    vp, ci, cf, cl = C.c_void_p, C.c_int, C.c_float, C.c_long
    fp = C.POINTER(C.c_float)
    sig = {
        "fvhd_version": (ci, []),
        "fvhd_last_error": (C.c_char_p, []),
        "fvhd_create": (ci, [C.POINTER(vp), ci, ci, ci]),
        "fvhd_destroy": (None, [vp]),
        "fvhd_reserve": (ci, [vp, ci]),
        "fvhd_set_tensor": (ci, [vp, C.c_char_p, vp, C.POINTER(C.c_int64), ci]),
        "fvhd_finalize_weights": (ci, [vp]),
        "fvhd_set_projector": (ci, [vp, vp, vp, vp, vp, ci, ci]),
        "fvhd_encode": (ci, [vp, vp, ci, ci, vp, ci, vp]),
        "fvhd_project": (ci, [vp, vp, ci, ci, vp, ci, vp]),
        "fvhd_encode_images": (ci, [vp, vp, ci, ci, vp, ci, vp]),
        "fvhd_num_steps": (ci, [vp]),
        "fvhd_step_info": (ci, [vp, ci] + [C.POINTER(ci)] * 7),
        "fvhd_run_steps": (ci, [vp, ci, ci, vp, ci, vp, vp]),
        "fvhd_num_tokens": (ci, [vp]),
        "fvhd_hidden_size": (ci, [vp]),
        "fvhd_profile_enable": (ci, [vp, ci]),
        "fvhd_profile_reset": (ci, [vp]),
        "fvhd_profile_read": (ci, [vp, ci, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(ci)]),
        "fvhd_op_dwconv": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci]),
        "fvhd_op_dw7_mfma": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci]),
        "fvhd_op_dw3_dw7": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]),
        "fvhd_dw3_dw7_supported": (ci, [ci, ci, ci, ci, ci]),
        "fvhd_dw7s2_mfma_supported": (ci, [ci, ci, ci, ci, ci]),
        "fvhd_op_dw7s2_mfma": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci]),
        "fvhd_op_gemm": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci]),
        "fvhd_op_gemm_splitk_ls": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci]),
        "fvhd_gemm_splitk_plan": (ci, [ci, ci, ci]),
        "fvhd_op_layernorm": (ci, [vp, vp, vp, vp, vp, ci, ci, cf]),
        "fvhd_op_attention": (ci, [vp, vp, vp, ci, ci, ci]),
        "fvhd_op_attention_fp8": (ci, [vp, vp, vp, ci, ci, ci]),
        "fvhd_set_attention_fp8": (ci, [vp, ci]),
        "fvhd_set_graph": (ci, [vp, ci]),
        "fvhd_set_batch_invariant": (ci, [vp, ci]),
        "fvhd_op_stem_conv": (ci, [vp, vp, ci, vp, vp, vp, ci, ci]),
        "fvhd_op_stem_fused": (ci, [vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci]),
        "fvhd_op_se_head": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci]),
        "fvhd_op_ffn_fused": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci]),
        "fvhd_ffn_fused_supported": (ci, [ci]),
        "fvhd_ffn_pack": (ci, [ci, vp, vp, vp, vp, ci]),
        "fvhd_set_ffn_precision": (ci, [vp, ci, ci]),
        "fvhd_get_ffn_precision": (ci, [vp, ci]),
        "fvhd_audit_ranges": (ci, [vp, vp, ci, ci, cf, C.POINTER(C.c_float), C.POINTER(ci), vp]),
        "fvhd_set_range_guard": (ci, [vp, ci]),
        "fvhd_range_guard_limit": (ci, [vp, ci, C.POINTER(C.c_float)]),
        "fvhd_range_guard_poll": (ci, [vp, ci, C.POINTER(ci), C.POINTER(C.c_float), ci, C.POINTER(ci)]),
        "fvhd_op_dw7_amax": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
        "fvhd_op_preprocess": (ci, [vp, vp, ci, ci, C.c_int64, ci, ci, C.c_uint32, vp, vp, ci, vp, vp, ci, ci, ci, vp, vp, ci, vp, ci]),
        "fvhd_op_splice": (ci, [vp] * 12 + [ci, ci, ci, ci, C.c_int64, C.c_int64, ci, ci]),
        "fvhd_llm_create": (ci, [C.POINTER(vp), ci, ci, ci, ci, ci, ci, ci, ci, cf, cf]),
        "fvhd_llm_destroy": (None, [vp]),
        "fvhd_llm_set_tensor": (ci, [vp, C.c_char_p, vp, ci, C.POINTER(C.c_int64), ci]),
        "fvhd_llm_finalize": (ci, [vp]),
        "fvhd_llm_reserve": (ci, [vp, ci, ci]),
        "fvhd_llm_prefill": (ci, [vp, vp, ci, vp, vp, ci, ci, vp, vp, vp, vp]),
        "fvhd_llm_debug_hidden": (ci, [vp, vp, ci, vp]),
        "fvhd_op_rmsnorm": (ci, [vp, vp, vp, vp, ci, ci, cf]),
        "fvhd_op_rope": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf]),
        "fvhd_gemm_qkv_rope_supported": (ci, [ci, ci, ci, ci, ci, ci]),
        "fvhd_op_gemm_qkv_rope": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf]),
        "fvhd_llm_set_tensor_device": (ci, [vp, C.c_char_p, vp, ci, C.POINTER(C.c_int64), ci, vp]),
        "fvhd_llm_set_max_positions": (ci, [vp, ci]),
        "fvhd_llm_workspace_generation": (ci, [vp]),
        "fvhd_op_attention_causal": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci]),
        "fvhd_op_gemm_splitk": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci]),
        "fvhd_op_qkv_splitk_rope": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, cf, ci]),
        "fvhd_op_gemm_splitk_norm": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, C.c_float]),
    }
    del fp, cl
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)     # AttributeError here = the .so does not export what fvhd.h declares
        fn.restype = res
        fn.argtypes = args


def load():
    """Returns the loaded library; raises if `libfvhd.so` has not been built
    (`python -m ml_fastvlm_amd.build` or `__graft_entry__.build()`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FvhdError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m ml_fastvlm_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback for this path.")
        lib = C.CDLL(LIB_PATH)
        _declare(lib)
        got = lib.fvhd_version()
        if got // 100 != ABI_VERSION // 100 or got < ABI_VERSION:
            raise FvhdError(f"{LIB_PATH} reports ABI version {got}, this binding was written for {ABI_VERSION} (include/fvhd.h FVHD_VERSION): "
                            "rebuild the library (`python -m ml_fastvlm_amd.build`) - argument lists differ between major versions")
        _lib = lib
    return _lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().fvhd_last_error()
        raise FvhdError(f"{what or 'libfvhd'} failed (code {code}): {msg.decode() if msg else ''}")


def dtype_code(dt) -> int:
    import torch
    try:
        return {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}[dt]
    except KeyError:
        raise FvhdError(f"unsupported dtype {dt}: the tower accepts float32, float16 and bfloat16") from None


def ptr(t) -> C.c_void_p:
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream_ptr(device) -> C.c_void_p:
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Context:
    """Owns one `fvhd_ctx` (packed weights + workspace on one device)."""

    def __init__(self, device_index: int, image_size: int, max_batch: int = 1):
        lib = load()
        h = C.c_void_p()
        check(lib.fvhd_create(C.byref(h), device_index, image_size, max_batch), "fvhd_create")
        self._h = h
        self.device_index = device_index
        self.image_size = image_size

    def close(self) -> None:
        if getattr(self, "_h", None):
            load().fvhd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, max_batch: int) -> None:
        """size the workspace for `max_batch` images now (synchronises; not allowed during stream capture)"""
        check(load().fvhd_reserve(self._h, int(max_batch)), "fvhd_reserve")

    def set_tensor(self, key: str, t) -> None:
        import torch
        t = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
        shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
        check(load().fvhd_set_tensor(self._h, key.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()), f"fvhd_set_tensor({key})")

    def set_tensors(self, tensors) -> None:
        """All floating-point tensors of a state dict in ONE device-to-host transfer (round 5: the per-tensor `.to("cpu")` of `set_tensor`
        was 629 small synchronising copies on every `.to()` / re-pack of the tower): same-dtype tensors are concatenated on their device,
        cast to fp32 there, copied once, and handed to `fvhd_set_tensor` as views of that one host buffer (the library copies them)."""
        import torch
        items = [(k, v.detach()) for k, v in tensors.items() if v.is_floating_point()]
        groups = {}
        for k, v in items:
            groups.setdefault((v.device, v.dtype), []).append((k, v))
        CHUNK = 32 << 20                                   # elements per transfer: the device-side concatenation (and its fp32 copy when the
        for (_dev, _dt), grp in groups.items():            # tensors are 16-bit) stay below ~200 MB however large the tower (round 6, advisor)
            i = 0
            while i < len(grp):
                j, n_el = i, 0
                while j < len(grp) and (j == i or n_el + grp[j][1].numel() <= CHUNK):
                    n_el += grp[j][1].numel()
                    j += 1
                part = grp[i:j]
                flat = torch.cat([v.reshape(-1) for _, v in part]).cpu().to(dtype=torch.float32).contiguous()   # cast on the host
                off = 0
                for k, v in part:
                    n = v.numel()
                    shape = (C.c_int64 * max(1, v.dim()))(*v.shape)
                    check(load().fvhd_set_tensor(self._h, k.encode(), C.c_void_p(flat.data_ptr() + 4 * off), shape, v.dim()), f"fvhd_set_tensor({k})")
                    off += n
                i = j

    def finalize(self) -> None:
        check(load().fvhd_finalize_weights(self._h), "fvhd_finalize_weights")

    def set_projector(self, w0, b0, w2, b2) -> None:
        import torch
        ts = [x.detach().to(device="cpu", dtype=torch.float32).contiguous() for x in (w0, b0, w2, b2)]
        hidden, mm_hidden = ts[0].shape
        check(load().fvhd_set_projector(self._h, *[C.c_void_p(x.data_ptr()) for x in ts], mm_hidden, hidden),
              "fvhd_set_projector")

    @property
    def num_tokens(self) -> int:
        return load().fvhd_num_tokens(self._h)

    def encode(self, images, out) -> None:
        check(load().fvhd_encode(self._h, ptr(images), dtype_code(images.dtype), images.shape[0], ptr(out),
                                 dtype_code(out.dtype), stream_ptr(images.device)), "fvhd_encode")

    def project(self, tokens, out) -> None:
        rows = tokens.numel() // tokens.shape[-1]
        check(load().fvhd_project(self._h, ptr(tokens), dtype_code(tokens.dtype), rows, ptr(out), dtype_code(out.dtype),
                                  stream_ptr(tokens.device)), "fvhd_project")

    def encode_images(self, images, out) -> None:
        check(load().fvhd_encode_images(self._h, ptr(images), dtype_code(images.dtype), images.shape[0], ptr(out),
                                        dtype_code(out.dtype), stream_ptr(images.device)), "fvhd_encode_images")

    # ---- step-level execution (tests) ----
    STEP_KINDS = ("stem", "cpe", "repmixer_block", "attention_block", "patch_embed", "conv_exp")

    def steps(self):
        """[(kind, stage, block, c_in, h_in, c_out, h_out)] in execution order."""
        lib, out = load(), []
        for i in range(lib.fvhd_num_steps(self._h)):
            v = [C.c_int(0) for _ in range(7)]
            check(lib.fvhd_step_info(self._h, i, *[C.byref(x) for x in v]), "fvhd_step_info")
            out.append((self.STEP_KINDS[v[0].value],) + tuple(x.value for x in v[1:]))
        return out

    def run_steps(self, first: int, last: int, x_in, x_out) -> None:
        check(load().fvhd_run_steps(self._h, first, last, ptr(x_in), x_in.shape[0], ptr(x_out), stream_ptr(x_in.device)),
              "fvhd_run_steps")

    def set_attention_fp8(self, on: bool) -> None:
        """e4m3 MFMA operands in the MHSA core (BASELINE.json configs[4], opt-in); default off = bf16 operands."""
        check(load().fvhd_set_attention_fp8(self._h, int(bool(on))), "fvhd_set_attention_fp8")

    def set_batch_invariant(self, on: bool) -> None:
        """kernel choice by image shape only: an image gives the same bits in any batch (default off = fastest kernel per batch size)."""
        check(load().fvhd_set_batch_invariant(self._h, int(bool(on))), "fvhd_set_batch_invariant")

    def ffn_precision(self, step: int) -> int:
        """FFN_HALF / FFN_BF16 of the fused ConvFFN of `step`, -1 for a step without one"""
        return load().fvhd_get_ffn_precision(self._h, int(step))

    def set_ffn_precision(self, step: int, precision: int) -> None:
        check(load().fvhd_set_ffn_precision(self._h, int(step), int(precision)), "fvhd_set_ffn_precision")

    def audit_ranges(self, images, switch_above: float = 65504.0):
        """One eager pass over `images`: -> (max |fc1 output| per step, number of blocks switched to FFN_BF16)."""
        n = load().fvhd_num_steps(self._h)
        out = (C.c_float * n)()
        sw = C.c_int(0)
        check(load().fvhd_audit_ranges(self._h, ptr(images), dtype_code(images.dtype), images.shape[0], float(switch_above), out, C.byref(sw),
                                       stream_ptr(images.device)), "fvhd_audit_ranges")
        return [out[i] for i in range(n)], sw.value

    def set_range_guard(self, on: bool) -> None:
        """the always-on range guard of the half-precision fused ConvFFN (include/fvhd.h "range guard"); default on"""
        check(load().fvhd_set_range_guard(self._h, int(bool(on))), "fvhd_set_range_guard")

    def range_guard_limit(self, step: int) -> float:
        """largest max |A| for which the fused block of `step` is provably inside the half-precision range"""
        out = C.c_float(0.0)
        check(load().fvhd_range_guard_limit(self._h, int(step), C.byref(out)), "fvhd_range_guard_limit")
        return out.value

    def range_guard_poll(self, wait: bool = False):
        """-> [(step, max |A|)] of the blocks the guard has moved to the bf16-operand form since the last poll"""
        cap = 64
        steps, amax, n = (C.c_int * cap)(), (C.c_float * cap)(), C.c_int(0)
        check(load().fvhd_range_guard_poll(self._h, int(bool(wait)), steps, amax, cap, C.byref(n)), "fvhd_range_guard_poll")
        return [(steps[i], amax[i]) for i in range(n.value)]

    def set_graph(self, on: bool) -> None:
        """replay the interior steps as one hipGraph per batch size (launch-bound small batches)."""
        check(load().fvhd_set_graph(self._h, int(bool(on))), "fvhd_set_graph")

    # ---- measurement ----
    def profile_enable(self, on: bool) -> None:
        check(load().fvhd_profile_enable(self._h, int(on)), "fvhd_profile_enable")

    def profile_reset(self) -> None:
        check(load().fvhd_profile_reset(self._h), "fvhd_profile_reset")

    def profile_read(self) -> Dict[str, Tuple[float, int]]:
        n = 32
        names = (C.c_char_p * n)()
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        got = C.c_int(0)
        check(load().fvhd_profile_read(self._h, n, names, ms, cnt, C.byref(got)), "fvhd_profile_read")
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(got.value)}
