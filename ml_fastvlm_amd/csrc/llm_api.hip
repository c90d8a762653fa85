// C ABI of the Qwen2 prefill (include/fvhd.h "LLM prefill"): context, weight packing, workspace and the launch sequence of
// transformers' Qwen2ForCausalLM.forward on inputs_embeds (the call the reference makes at llava/model/language_model/
// llava_qwen.py:92-103 after prepare_inputs_labels_for_multimodal, and from generate(), :138-143).  Kernels: llm.hip + gemm.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/fvhd.h"

extern "C" {
int fvhd_launch_gemm(hipStream_t, const void*, const void*, const float*, const float*, const void*, void*, int, int, int, int, int);
int fvhd_launch_gemm_splitk(hipStream_t, const void*, const void*, const void*, void*, float*, int, int, int, int);
int fvhd_launch_gemm_splitk_partials(hipStream_t, const void*, const void*, float*, int, int, int, int);
int fvhd_launch_splitk_bias_rope(hipStream_t, const float*, int, int, const float*, void*, const long*, const float*, void*, void*, int, int, int, int, int, int, float);
int fvhd_launch_gemm_splitk_norm(hipStream_t, const void*, const void*, const void*, void*, float*, int, int, int, int, const float*, void*, float);
int fvhd_launch_rmsnorm(hipStream_t, const void*, void*, const float*, int, int, float);
int fvhd_launch_rope(hipStream_t, void*, const long*, const float*, void*, void*, int, int, int, int, int, int, float);
int fvhd_gemm_qkv_rope_supported(int, int, int, int, int, int);
int fvhd_launch_gemm_qkv_rope(hipStream_t, const void*, const void*, const float*, void*, int, int, int, const long*, const float*, void*, void*, int, int, int, int, int, int, float);
int fvhd_launch_llm_attention(hipStream_t, const void*, void*, const unsigned char*, int, int, int, int, int);
int fvhd_launch_cast_rows(hipStream_t, const void*, int, void*, long);
int fvhd_launch_gather_rows(hipStream_t, const void*, void*, int, int, int, int);
int fvhd_set_error(const char* msg);     // fvhd_api.hip: the library's one thread-local error string
}

namespace {

constexpr int EPI_NONE = 0, EPI_BIAS = 1, EPI_RESID = 4, EPI_SWIGLU = 5;
constexpr int kMaxSplits = 4;

int lfail(const std::string& m) { return fvhd_set_error(m.c_str()); }
int lhip(const char* what, hipError_t e) { return lfail(std::string(what) + ": " + hipGetErrorString(e)); }

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

uint16_t bf16_rne(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

float half_to_float(uint16_t h)
{
    const uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 31, m = h & 1023;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = s;
        else { int k = 0; uint32_t mm = m; while (!(mm & 1024)) { mm <<= 1; ++k; } u = s | ((uint32_t)(113 - k) << 23) | ((mm & 1023) << 13); }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

float load_as_float(const void* p, int dtype, size_t i)
{
    if (dtype == FVHD_F32) return ((const float*)p)[i];
    if (dtype == FVHD_BF16) { uint32_t u = (uint32_t)((const uint16_t*)p)[i] << 16; float f; memcpy(&f, &u, 4); return f; }
    return half_to_float(((const uint16_t*)p)[i]);
}

struct LayerOff { size_t ln1, wqkv, bqkv, wo, ln2, wgu, wd; };

struct DevGuard {
    int prev = -1; bool sw = false; hipError_t err = hipSuccess;
    explicit DevGuard(int d) { err = hipGetDevice(&prev); if (err == hipSuccess && prev != d) { err = hipSetDevice(d); sw = err == hipSuccess; } }
    ~DevGuard() { if (sw) (void)hipSetDevice(prev); }
};

}  // namespace

struct fvhd_llm {
    int device = 0, H = 0, L = 0, nh = 0, nkv = 0, hd = 0, I = 0, V = 0;
    float eps = 1e-6f, theta = 1e6f;
    int qkvw = 0;
    char* wdev = nullptr;
    size_t wbytes = 0;
    std::vector<LayerOff> lo;
    size_t norm_off = 0, lm_off = 0;
    std::vector<char> got;                 // per expected tensor: received?
    std::vector<std::string> names;
    // workspace
    char* ws = nullptr;
    size_t ws_bytes = 0;
    int ws_rows = 0, ws_batch = 0, ws_pos = 0;
    char *h = nullptr, *xn = nullptr, *qkv = nullptr, *att = nullptr, *act = nullptr, *last = nullptr, *lastn = nullptr;
    float *rope = nullptr, *part = nullptr;
    int fuse_rope = 0;                     // FVHD_LLM_FUSEROPE=1: rotary embedding + KV-cache copies inside the q|k|v projection's epilogue instead of their own launch (identical bits; measured neutral - prefill 3.421 / 3.410 -> 3.404 / 3.407 ms at B = 8, 2.316 -> 2.342 at B = 1, profiles/r05_ttft_fuserope_ab.log: the 5.4-us launch saved comes back as epilogue time - so off by default)
    int down_splits = kMaxSplits, o_splits = 2, qkv_splits = 0, fuse_norm = 1;     // FVHD_LLM_SPLITK / FVHD_LLM_OSPLIT (largest split of down_proj / o_proj, 0 = never) / FVHD_LLM_QKVSPLIT / FVHD_LLM_FUSENORM
    int max_pos = 0;                       // fvhd_llm_set_max_positions (config.max_position_embeddings): rows of the rotary table
    // A prefill that ran while its stream was being captured put this workspace's pointers into the CALLER's graph.  Such a workspace is
    // never freed when a later call needs a bigger one: it is retired (kept until fvhd_llm_destroy), so the captured graph keeps
    // replaying on valid memory.  `generation` counts workspace replacements (fvhd_llm_workspace_generation).
    bool ws_captured = false;
    std::vector<char*> retired;
    int generation = 0;
    // fvhd_llm_set_tensor_device enqueues its copies on the CALLER's stream: `load_ev` is recorded behind the latest one so that
    // fvhd_llm_finalize (host wait) and fvhd_llm_prefill (stream wait, whatever stream it runs on) are ordered after the packing
    hipEvent_t load_ev = nullptr;
    hipStream_t load_stream = nullptr;
    bool load_pending = false;
};

namespace {

// expected tensor index: per layer 12 (ln1, q.w, q.b, k.w, k.b, v.w, v.b, o.w, ln2, gate, up, down), then norm, lm_head
int tensor_index(const fvhd_llm* c, const std::string& key, int* layer, int* which)
{
    std::string k = key;
    if (k.rfind("model.", 0) == 0) k = k.substr(6);
    if (k == "norm.weight") { *layer = -1; *which = 0; return c->L * 12; }
    if (k == "lm_head.weight") { *layer = -1; *which = 1; return c->L * 12 + 1; }
    if (k.rfind("layers.", 0) != 0) return -1;
    const size_t dot = k.find('.', 7);
    if (dot == std::string::npos) return -1;
    const int l = atoi(k.substr(7, dot - 7).c_str());
    if (l < 0 || l >= c->L) return -1;
    const std::string rest = k.substr(dot + 1);
    static const char* kNames[12] = {"input_layernorm.weight", "self_attn.q_proj.weight", "self_attn.q_proj.bias", "self_attn.k_proj.weight",
                                     "self_attn.k_proj.bias", "self_attn.v_proj.weight", "self_attn.v_proj.bias", "self_attn.o_proj.weight",
                                     "post_attention_layernorm.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight"};
    for (int i = 0; i < 12; ++i)
        if (rest == kNames[i]) { *layer = l; *which = i; return l * 12 + i; }
    return -1;
}

// host rows [rows][cols] of `dtype` -> device bf16 rows at dst, dst row pitch `pitch_elems` (interleaving = pitch 2 * cols)
int upload_matrix(const void* host, int dtype, size_t rows, size_t cols, char* dst, size_t pitch_elems)
{
    std::vector<uint16_t> tmp(rows * cols);
    if (dtype == FVHD_BF16) memcpy(tmp.data(), host, rows * cols * 2);
    else
        for (size_t i = 0; i < rows * cols; ++i) tmp[i] = bf16_rne(load_as_float(host, dtype, i));
    hipError_t e = hipMemcpy2D(dst, pitch_elems * 2, tmp.data(), cols * 2, cols * 2, rows, hipMemcpyHostToDevice);
    return e == hipSuccess ? 0 : lhip("hipMemcpy2D(llm weights)", e);
}

int upload_vector_f32(const void* host, int dtype, size_t n, char* dst)
{
    std::vector<float> tmp(n);
    for (size_t i = 0; i < n; ++i) tmp[i] = load_as_float(host, dtype, i);
    hipError_t e = hipMemcpy(dst, tmp.data(), n * 4, hipMemcpyHostToDevice);
    return e == hipSuccess ? 0 : lhip("hipMemcpy(llm vector)", e);
}

int ensure_ws(fvhd_llm* c, int B, int T, hipStream_t st, bool check_capture)
{
    const int rows = (int)(((size_t)B * T + 255) / 256 * 256);
    if (c->ws && rows <= c->ws_rows && B <= c->ws_batch && T <= c->ws_pos) return 0;      // (ws_pos >= 8192 once allocated)
    if (check_capture) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return lfail("fvhd_llm_prefill: the workspace must grow for this (batch, length) but the stream is being captured - call fvhd_llm_reserve first");
    }
    // the rotary table covers max_position_embeddings (fvhd_llm_set_max_positions; at most 65536 rows = 16 MB at head_dim 64) or 8192
    // positions: position ids of a prefill are < seq_len, and a caller continuing a longer context may pass larger ones - beyond the
    // table the kernel computes the phases itself (llm.hip: rope_kernel), it never clamps
    const int want_pos = c->max_pos > 0 ? (c->max_pos < 65536 ? c->max_pos : 65536) : 8192;
    const int tpos = T > want_pos ? T : want_pos;
    const int nrows = rows > c->ws_rows ? rows : c->ws_rows, nb = B > c->ws_batch ? B : c->ws_batch, np = tpos > c->ws_pos ? tpos : c->ws_pos;
    const int lb = (nb + 15) / 16 * 16;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al256(bytes); return o; };
    const size_t o_h = take((size_t)nrows * c->H * 2), o_xn = take((size_t)nrows * c->H * 2), o_qkv = take((size_t)nrows * c->qkvw * 2),
                 o_att = take((size_t)nrows * c->nh * c->hd * 2), o_act = take((size_t)nrows * c->I * 2), o_last = take((size_t)lb * c->H * 2),
                 o_lastn = take((size_t)lb * c->H * 2), o_rope = take((size_t)np * c->hd * 4),
                 o_part = take((size_t)nrows * std::max(kMaxSplits * c->H, 2 * c->qkvw) * 4);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return lhip("hipDeviceSynchronize", e);
    if (c->ws) {
        if (c->ws_captured) c->retired.push_back(c->ws);      // a caller's graph may still replay on it
        else (void)hipFree(c->ws);
    }
    c->ws = nullptr;
    c->ws_captured = false;
    ++c->generation;
    e = hipMalloc((void**)&c->ws, off);
    if (e != hipSuccess) return lhip("hipMalloc(llm workspace)", e);
    e = hipMemset(c->ws, 0, off);           // the padding rows start (and stay) finite
    if (e != hipSuccess) return lhip("hipMemset(llm workspace)", e);
    c->ws_bytes = off; c->ws_rows = nrows; c->ws_batch = nb; c->ws_pos = np;
    c->h = c->ws + o_h; c->xn = c->ws + o_xn; c->qkv = c->ws + o_qkv; c->att = c->ws + o_att; c->act = c->ws + o_act;
    c->last = c->ws + o_last; c->lastn = c->ws + o_lastn; c->rope = (float*)(c->ws + o_rope); c->part = (float*)(c->ws + o_part);
    // rotary table, fp32 like Qwen2RotaryEmbedding.forward: inv_freq_i = theta^(-2i/hd), angle = pos * inv_freq_i, (cos, sin)
    std::vector<float> tab((size_t)np * c->hd);
    for (int p = 0; p < np; ++p)
        for (int i = 0; i < c->hd / 2; ++i) {
            const float inv = 1.0f / powf(c->theta, (float)(2 * i) / (float)c->hd);
            const float ang = (float)p * inv;
            tab[((size_t)p * (c->hd / 2) + i) * 2] = cosf(ang);
            tab[((size_t)p * (c->hd / 2) + i) * 2 + 1] = sinf(ang);
        }
    e = hipMemcpy(c->rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    return e == hipSuccess ? 0 : lhip("hipMemcpy(rope table)", e);
}

#define LCHECK(expr, what)                                   \
    do {                                                     \
        int _e = (expr);                                     \
        if (_e) return lhip(what, (hipError_t)_e);           \
    } while (0)

}  // namespace

extern "C" {

int fvhd_llm_create(fvhd_llm** out, int device, int hidden, int n_layers, int n_heads, int n_kv_heads, int head_dim, int intermediate,
                    int vocab, float rms_eps, float rope_theta)
{
    if (!out) return lfail("fvhd_llm_create: out is NULL");
    *out = nullptr;
    if (hidden <= 0 || n_layers <= 0 || n_heads <= 0 || n_kv_heads <= 0 || intermediate <= 0 || vocab <= 0)
        return lfail("fvhd_llm_create: sizes must be positive");
    if (head_dim != 64 && head_dim != 128) return lfail("fvhd_llm_create: head_dim must be 64 or 128 (Qwen2-0.5B/1.5B: 64 / 128, 7B: 128)");
    if (n_heads % n_kv_heads) return lfail("fvhd_llm_create: n_heads must be a multiple of n_kv_heads");
    const int qkvw = (n_heads + 2 * n_kv_heads) * head_dim;
    if (hidden % 64 || (n_heads * head_dim) % 64 || intermediate % 64 || qkvw % 16 || vocab % 16)
        return lfail("fvhd_llm_create: hidden, n_heads * head_dim and intermediate must be multiples of 64, vocab of 16");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return lfail("fvhd_llm_create: no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n) return lfail("fvhd_llm_create: bad device index");
    fvhd_llm* c = new fvhd_llm();
    c->device = device; c->H = hidden; c->L = n_layers; c->nh = n_heads; c->nkv = n_kv_heads; c->hd = head_dim; c->I = intermediate; c->V = vocab;
    c->eps = rms_eps; c->theta = rope_theta; c->qkvw = qkvw;
    if (const char* ev = getenv("FVHD_LLM_SPLITK")) c->down_splits = atoi(ev);      // 0 = never split (A/B)
    if (const char* ev = getenv("FVHD_LLM_OSPLIT")) c->o_splits = atoi(ev);
    if (const char* ev = getenv("FVHD_LLM_QKVSPLIT")) c->qkv_splits = atoi(ev);
    if (const char* ev = getenv("FVHD_LLM_FUSENORM")) c->fuse_norm = atoi(ev);
    if (const char* ev = getenv("FVHD_LLM_FUSEROPE")) c->fuse_rope = atoi(ev);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al256(bytes); return o; };
    c->lo.resize(n_layers);
    for (int l = 0; l < n_layers; ++l) {
        LayerOff& o = c->lo[l];
        o.ln1 = take((size_t)hidden * 4);
        o.wqkv = take((size_t)qkvw * hidden * 2);
        o.bqkv = take((size_t)qkvw * 4);
        o.wo = take((size_t)hidden * n_heads * head_dim * 2);
        o.ln2 = take((size_t)hidden * 4);
        o.wgu = take((size_t)2 * intermediate * hidden * 2);
        o.wd = take((size_t)hidden * intermediate * 2);
    }
    c->norm_off = take((size_t)hidden * 4);
    c->lm_off = take((size_t)vocab * hidden * 2);
    c->wbytes = off;
    c->got.assign((size_t)n_layers * 12 + 2, 0);
    DevGuard g(device);
    if (g.err != hipSuccess) { delete c; return lhip("hipSetDevice", g.err); }
    hipError_t e = hipMalloc((void**)&c->wdev, off);
    if (e != hipSuccess) { delete c; return lhip("hipMalloc(llm weights)", e); }
    *out = c;
    return 0;
}

void fvhd_llm_destroy(fvhd_llm* c)
{
    if (!c) return;
    DevGuard g(c->device);
    (void)hipDeviceSynchronize();
    if (c->wdev) (void)hipFree(c->wdev);
    if (c->ws) (void)hipFree(c->ws);
    for (char* p : c->retired) (void)hipFree(p);
    if (c->load_ev) (void)hipEventDestroy(c->load_ev);
    delete c;
}

int fvhd_llm_set_tensor(fvhd_llm* c, const char* key, const void* host_data, int dtype, const int64_t* shape, int ndim)
{
    if (!c || !key || !host_data || !shape) return lfail("fvhd_llm_set_tensor: NULL argument");
    if (dtype < 0 || dtype > 2) return lfail("fvhd_llm_set_tensor: bad dtype");
    int layer = -1, which = -1;
    const int idx = tensor_index(c, key, &layer, &which);
    if (idx < 0) return lfail(std::string("fvhd_llm_set_tensor: not a tensor of the Qwen2 decoder stack: ") + key);
    const int H = c->H, hd = c->hd, nh = c->nh, nkv = c->nkv, I = c->I;
    auto want = [&](int64_t r, int64_t cc) -> bool { return cc < 0 ? (ndim == 1 && shape[0] == r) : (ndim == 2 && shape[0] == r && shape[1] == cc); };
    DevGuard g(c->device);
    if (g.err != hipSuccess) return lhip("hipSetDevice", g.err);
    int e = 0;
    bool ok = true;
    if (layer < 0) {
        if (which == 0) { ok = want(H, -1); if (ok) e = upload_vector_f32(host_data, dtype, H, c->wdev + c->norm_off); }
        else { ok = want(c->V, H); if (ok) e = upload_matrix(host_data, dtype, c->V, H, c->wdev + c->lm_off, H); }
    } else {
        const LayerOff& o = c->lo[layer];
        char* w = c->wdev;
        switch (which) {
        case 0: ok = want(H, -1); if (ok) e = upload_vector_f32(host_data, dtype, H, w + o.ln1); break;
        case 1: ok = want((int64_t)nh * hd, H); if (ok) e = upload_matrix(host_data, dtype, (size_t)nh * hd, H, w + o.wqkv, H); break;
        case 2: ok = want((int64_t)nh * hd, -1); if (ok) e = upload_vector_f32(host_data, dtype, (size_t)nh * hd, w + o.bqkv); break;
        case 3: ok = want((int64_t)nkv * hd, H); if (ok) e = upload_matrix(host_data, dtype, (size_t)nkv * hd, H, w + o.wqkv + (size_t)nh * hd * H * 2, H); break;
        case 4: ok = want((int64_t)nkv * hd, -1); if (ok) e = upload_vector_f32(host_data, dtype, (size_t)nkv * hd, w + o.bqkv + (size_t)nh * hd * 4); break;
        case 5: ok = want((int64_t)nkv * hd, H); if (ok) e = upload_matrix(host_data, dtype, (size_t)nkv * hd, H, w + o.wqkv + (size_t)(nh + nkv) * hd * H * 2, H); break;
        case 6: ok = want((int64_t)nkv * hd, -1); if (ok) e = upload_vector_f32(host_data, dtype, (size_t)nkv * hd, w + o.bqkv + (size_t)(nh + nkv) * hd * 4); break;
        case 7: ok = want(H, (int64_t)nh * hd); if (ok) e = upload_matrix(host_data, dtype, H, (size_t)nh * hd, w + o.wo, (size_t)nh * hd); break;
        case 8: ok = want(H, -1); if (ok) e = upload_vector_f32(host_data, dtype, H, w + o.ln2); break;
        // gate / up rows interleaved (row 2j = gate_j, row 2j + 1 = up_j): the SwiGLU epilogue of the GEMM pairs adjacent columns
        case 9: ok = want(I, H); if (ok) e = upload_matrix(host_data, dtype, I, H, w + o.wgu, (size_t)2 * H); break;
        case 10: ok = want(I, H); if (ok) e = upload_matrix(host_data, dtype, I, H, w + o.wgu + (size_t)H * 2, (size_t)2 * H); break;
        case 11: ok = want(H, I); if (ok) e = upload_matrix(host_data, dtype, H, I, w + o.wd, I); break;
        }
    }
    if (!ok) return lfail(std::string("fvhd_llm_set_tensor: bad shape for ") + key);
    if (e) return e;
    c->got[idx] = 1;
    return 0;
}

// The same tensors from DEVICE memory (a model that already lives on the GPU): matrices bf16, vectors fp32, row-major contiguous, on
// the context's device.  One device-to-device (2-D) copy per tensor on `stream` - no host round trip (advisor, round 3: from_hf moved
// 15 GB of a 7B model through the CPU).  The caller keeps `dev_data` alive until the stream has run the copy.
int fvhd_llm_set_tensor_device(fvhd_llm* c, const char* key, const void* dev_data, int dtype, const int64_t* shape, int ndim, fvhd_stream_t stream)
{
    if (!c || !key || !dev_data || !shape) return lfail("fvhd_llm_set_tensor_device: NULL argument");
    int layer = -1, which = -1;
    const int idx = tensor_index(c, key, &layer, &which);
    if (idx < 0) return lfail(std::string("fvhd_llm_set_tensor_device: not a tensor of the Qwen2 decoder stack: ") + key);
    const size_t H = c->H, hd = c->hd, nh = c->nh, nkv = c->nkv, I = c->I;
    // destination (offset, rows, cols, row pitch in elements) of every slot; vectors: cols = 0
    size_t off = 0, rows = 0, cols = 0, pitch = 0;
    if (layer < 0) {
        if (which == 0) { off = c->norm_off; rows = H; }
        else { off = c->lm_off; rows = c->V; cols = H; pitch = H; }
    } else {
        const LayerOff& o = c->lo[layer];
        switch (which) {
        case 0: off = o.ln1; rows = H; break;
        case 1: off = o.wqkv; rows = nh * hd; cols = H; pitch = H; break;
        case 2: off = o.bqkv; rows = nh * hd; break;
        case 3: off = o.wqkv + nh * hd * H * 2; rows = nkv * hd; cols = H; pitch = H; break;
        case 4: off = o.bqkv + nh * hd * 4; rows = nkv * hd; break;
        case 5: off = o.wqkv + (nh + nkv) * hd * H * 2; rows = nkv * hd; cols = H; pitch = H; break;
        case 6: off = o.bqkv + (nh + nkv) * hd * 4; rows = nkv * hd; break;
        case 7: off = o.wo; rows = H; cols = nh * hd; pitch = nh * hd; break;
        case 8: off = o.ln2; rows = H; break;
        case 9: off = o.wgu; rows = I; cols = H; pitch = 2 * H; break;              // gate rows at even, up rows at odd positions
        case 10: off = o.wgu + H * 2; rows = I; cols = H; pitch = 2 * H; break;
        case 11: off = o.wd; rows = H; cols = I; pitch = I; break;
        }
    }
    const bool vec = cols == 0;
    if (vec ? !(ndim == 1 && (size_t)shape[0] == rows) : !(ndim == 2 && (size_t)shape[0] == rows && (size_t)shape[1] == cols))
        return lfail(std::string("fvhd_llm_set_tensor_device: bad shape for ") + key);
    if (dtype != (vec ? FVHD_F32 : FVHD_BF16))
        return lfail(std::string("fvhd_llm_set_tensor_device: matrices must be bf16 and vectors fp32 on the device (") + key + ")");
    DevGuard g(c->device);
    if (g.err != hipSuccess) return lhip("hipSetDevice", g.err);
    hipError_t e = vec ? hipMemcpyAsync(c->wdev + off, dev_data, rows * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream)
                       : hipMemcpy2DAsync(c->wdev + off, pitch * 2, dev_data, cols * 2, cols * 2, rows, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return lhip("hipMemcpyAsync(llm weights, device to device)", e);
    // order later work after this copy (advisor, round 4: a prefill on ANOTHER stream could read half-packed weights)
    if (!c->load_ev && (e = hipEventCreateWithFlags(&c->load_ev, hipEventDisableTiming)) != hipSuccess) return lhip("hipEventCreate", e);
    if (c->load_pending && c->load_stream != (hipStream_t)stream) (void)hipEventSynchronize(c->load_ev);   // copies on a second stream: the event follows one stream at a time
    if ((e = hipEventRecord(c->load_ev, (hipStream_t)stream)) != hipSuccess) return lhip("hipEventRecord", e);
    c->load_stream = (hipStream_t)stream;
    c->load_pending = true;
    c->got[idx] = 1;
    return 0;
}

// every copy fvhd_llm_set_tensor_device has enqueued so far has completed (host wait); the caller holds a DevGuard
static int wait_for_loads(fvhd_llm* c)
{
    if (!c->load_pending) return 0;
    const hipError_t e = hipEventSynchronize(c->load_ev);
    if (e != hipSuccess) return lhip("hipEventSynchronize(llm weights)", e);
    c->load_pending = false;
    return 0;
}

int fvhd_llm_set_max_positions(fvhd_llm* c, int max_position_embeddings)
{
    if (!c || max_position_embeddings <= 0) return lfail("fvhd_llm_set_max_positions: bad argument");
    c->max_pos = max_position_embeddings;      // takes effect at the next workspace (re)allocation: call it before fvhd_llm_reserve
    return 0;
}

int fvhd_llm_workspace_generation(const fvhd_llm* c) { return c ? c->generation : -1; }

int fvhd_llm_finalize(fvhd_llm* c)
{
    if (!c) return lfail("fvhd_llm_finalize: ctx is NULL");
    for (size_t i = 0; i < c->got.size(); ++i)
        if (!c->got[i]) {
            const size_t l = i / 12, w = i % 12;
            return lfail("fvhd_llm_finalize: missing tensor (layer " + std::to_string(l) + ", slot " + std::to_string(w) +
                         "; slots: ln1 q.w q.b k.w k.b v.w v.b o.w ln2 gate up down | norm lm_head)");
        }
    DevGuard g(c->device);
    if (g.err != hipSuccess) return lhip("hipSetDevice", g.err);
    return wait_for_loads(c);               // the device-to-device packing is complete when this returns (include/fvhd.h "stream contract")
}

int fvhd_llm_reserve(fvhd_llm* c, int batch, int seq_len)
{
    if (!c || batch <= 0 || seq_len <= 0) return lfail("fvhd_llm_reserve: bad argument");
    DevGuard g(c->device);
    if (g.err != hipSuccess) return lhip("hipSetDevice", g.err);
    return ensure_ws(c, batch, seq_len, nullptr, false);
}

int fvhd_llm_prefill(fvhd_llm* c, const void* embeds, int dtype, const uint8_t* key_valid, const int64_t* position_ids, int batch, int seq_len,
                     float* logits_out, void* k_cache, void* v_cache, fvhd_stream_t stream)
{
    if (!c || !embeds || !logits_out) return lfail("fvhd_llm_prefill: NULL argument");
    if (dtype < 0 || dtype > 2) return lfail("fvhd_llm_prefill: bad dtype");
    if (batch <= 0 || seq_len <= 0) return lfail("fvhd_llm_prefill: batch and seq_len must be positive");
    if ((k_cache == nullptr) != (v_cache == nullptr)) return lfail("fvhd_llm_prefill: k_cache and v_cache come together");
    for (char g : c->got)
        if (!g) return lfail("fvhd_llm_prefill: weights incomplete (fvhd_llm_finalize reports the missing tensor)");
    DevGuard g(c->device);
    if (g.err != hipSuccess) return lhip("hipSetDevice", g.err);
    hipStream_t st = (hipStream_t)stream;
    int e = ensure_ws(c, batch, seq_len, st, true);
    if (e) return e;
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        if (capturing) c->ws_captured = true;
        // tensors re-set after fvhd_llm_finalize: this prefill runs behind their copies (a captured stream cannot wait on an outside
        // event - there the host waits once)
        if (c->load_pending) {
            // hipEventQuery / hipEventSynchronize are not capture-safe under the default (global) capture mode: while the caller's stream is
            // capturing they run in relaxed mode, so that they cannot invalidate the caller's capture (round 6, advisor)
            hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
            if (capturing) (void)hipThreadExchangeStreamCaptureMode(&mode);
            const hipError_t q = hipEventQuery(c->load_ev);
            if (q == hipSuccess) c->load_pending = false;
            else if (capturing) e = wait_for_loads(c);
            if (capturing) (void)hipThreadExchangeStreamCaptureMode(&mode);
            if (q != hipSuccess && q != hipErrorNotReady) (void)hipGetLastError();
            if (e) return e;
            if (c->load_pending && !capturing && st != c->load_stream) {
                const hipError_t he = hipStreamWaitEvent(st, c->load_ev, 0);
                if (he != hipSuccess) return lhip("hipStreamWaitEvent(llm weights)", he);
            }
        }
    }
    const int B = batch, T = seq_len, M = B * T, Mp = (M + 255) / 256 * 256;
    const int H = c->H, I = c->I, nh = c->nh, nkv = c->nkv, hd = c->hd;
    const char* w = c->wdev;
    if ((size_t)M * H % 4) return lfail("fvhd_llm_prefill: batch * seq_len * hidden must be a multiple of 4");
    LCHECK(fvhd_launch_cast_rows(st, embeds, dtype, c->h, (long)M * H), "cast embeds");
    const size_t cache_layer = (size_t)B * nkv * T * hd * 2;
    // split-K factor of a GEMM with few output tiles (Mp / 128 x H / 128) - while the tiles of one slice do not fill the chip twice over, K is
    // split across workgroups (fp32 partials + a deterministic reduce that also adds the residual): down_proj 70 -> ~25 us per layer at the
    // 0.5 B prefill shape (B = 8 x 285 tokens).  The reduce of a split GEMM also applies the RMSNorm the next operation starts with
    // (splitk_reduce_norm_kernel, bit-identical to the separate launch): input_layernorm of layer l + 1 behind down_proj of layer l, and -
    // when o_proj is split too (FVHD_LLM_OSPLIT) - post_attention_layernorm behind o_proj
    // the largest split that still fits ONE round of the streaming 128 x 128 kernel (<= 256 workgroups: gemm.hip v1s) - else, as in round 3, the
    // largest within two v1 workgroups per CU.  0.5 B at B = 8 (126 tiles): down_proj in TWO slices of 38 K steps on v1s instead of four of
    // 19 on v1, 33 -> 16 MB of partials: prefill 3.92 -> 3.77 ms (profiles/r04_ttft_down_split.log)
    int ncu_ = 0;
    if (hipDeviceGetAttribute(&ncu_, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || ncu_ <= 0) ncu_ = 256;
    const long ncu = ncu_;
    auto pick_splits = [&](int N, int K, int max_sp) {
        const long tiles = (long)(Mp / 128) * (N / 128);
        if (N % 128 == 0)
            for (long cap = ncu; cap <= 2 * ncu; cap += ncu)        // one round, then two rounds, of one workgroup per CU (256 / 512 on MI355X)
                for (int sp = kMaxSplits; sp > 1; sp >>= 1)
                    if (sp <= max_sp && tiles * sp <= cap && K % (64 * sp) == 0) return sp;
        return 1;
    };
    const int down_sp = pick_splits(H, I, c->down_splits), o_sp = pick_splits(H, nh * hd, c->o_splits);
    // q|k|v projection: split in two, the reduce applies bias + rotary embedding + the KV-cache copies (splitk_bias_rope_kernel)
    const int qkv_sp = pick_splits(c->qkvw, H, std::min(c->qkv_splits, 2));
    bool xn_ready = false;                      // c->xn already holds input_layernorm(c->h) of the coming layer
    for (int l = 0; l < c->L; ++l) {
        const LayerOff& o = c->lo[l];
        if (!xn_ready) LCHECK(fvhd_launch_rmsnorm(st, c->h, c->xn, (const float*)(w + o.ln1), Mp, H, c->eps), "rmsnorm 1");
        void* kc = k_cache ? (char*)k_cache + l * cache_layer : nullptr;
        void* vc = v_cache ? (char*)v_cache + l * cache_layer : nullptr;
        if (qkv_sp > 1) {
            LCHECK(fvhd_launch_gemm_splitk_partials(st, c->xn, w + o.wqkv, c->part, Mp, c->qkvw, H, qkv_sp), "qkv gemm (split-K)");
            LCHECK(fvhd_launch_splitk_bias_rope(st, c->part, qkv_sp, Mp, (const float*)(w + o.bqkv), c->qkv, (const long*)position_ids, c->rope, kc, vc,
                                                M, T, nh, nkv, hd, c->ws_pos, c->theta), "qkv reduce + bias + rope");
        } else if (c->fuse_rope && fvhd_gemm_qkv_rope_supported(Mp, c->qkvw, H, hd, nh, nkv)) {
            // round 5: bias + rotary embedding + the KV-cache copies in the projection's own epilogue (head_dim 64; bit-identical to the two launches)
            LCHECK(fvhd_launch_gemm_qkv_rope(st, c->xn, w + o.wqkv, (const float*)(w + o.bqkv), c->qkv, Mp, c->qkvw, H, (const long*)position_ids, c->rope, kc, vc,
                                             M, T, nh, nkv, hd, c->ws_pos, c->theta), "qkv gemm + rope");
        } else {
            LCHECK(fvhd_launch_gemm(st, c->xn, w + o.wqkv, (const float*)(w + o.bqkv), nullptr, nullptr, c->qkv, Mp, c->qkvw, H, EPI_BIAS, FVHD_BF16), "qkv gemm");
            LCHECK(fvhd_launch_rope(st, c->qkv, (const long*)position_ids, c->rope, kc, vc, M, T, nh, nkv, hd, c->ws_pos, c->theta), "rope");
        }
        LCHECK(fvhd_launch_llm_attention(st, c->qkv, c->att, key_valid, B, T, nh, nkv, hd), "attention");
        if (o_sp > 1 && c->fuse_norm) {
            LCHECK(fvhd_launch_gemm_splitk_norm(st, c->att, w + o.wo, c->h, c->h, c->part, Mp, H, nh * hd, o_sp, (const float*)(w + o.ln2), c->xn, c->eps),
                   "o_proj gemm (split-K + rmsnorm 2)");
        } else {
            if (o_sp > 1) LCHECK(fvhd_launch_gemm_splitk(st, c->att, w + o.wo, c->h, c->h, c->part, Mp, H, nh * hd, o_sp), "o_proj gemm (split-K)");
            else LCHECK(fvhd_launch_gemm(st, c->att, w + o.wo, nullptr, nullptr, c->h, c->h, Mp, H, nh * hd, EPI_RESID, FVHD_BF16), "o_proj gemm");
            LCHECK(fvhd_launch_rmsnorm(st, c->h, c->xn, (const float*)(w + o.ln2), Mp, H, c->eps), "rmsnorm 2");
        }
        LCHECK(fvhd_launch_gemm(st, c->xn, w + o.wgu, nullptr, nullptr, nullptr, c->act, Mp, 2 * I, H, EPI_SWIGLU, FVHD_BF16), "gate_up gemm");
        xn_ready = false;
        if (down_sp > 1 && c->fuse_norm && l + 1 < c->L) {
            LCHECK(fvhd_launch_gemm_splitk_norm(st, c->act, w + o.wd, c->h, c->h, c->part, Mp, H, I, down_sp, (const float*)(w + c->lo[l + 1].ln1), c->xn, c->eps),
                   "down gemm (split-K + rmsnorm 1 of the next layer)");
            xn_ready = true;
        } else if (down_sp > 1) {
            LCHECK(fvhd_launch_gemm_splitk(st, c->act, w + o.wd, c->h, c->h, c->part, Mp, H, I, down_sp), "down gemm (split-K)");
        } else {
            LCHECK(fvhd_launch_gemm(st, c->act, w + o.wd, nullptr, nullptr, c->h, c->h, Mp, H, I, EPI_RESID, FVHD_BF16), "down gemm");
        }
    }
    // logits of the LAST position of every sequence (what generate() reads: outputs.logits[:, -1, :])
    LCHECK(fvhd_launch_gather_rows(st, c->h, c->last, B, T, T - 1, H), "gather last rows");
    LCHECK(fvhd_launch_rmsnorm(st, c->last, c->lastn, (const float*)(w + c->norm_off), B, H, c->eps), "final norm");
    LCHECK(fvhd_launch_gemm(st, c->lastn, w + c->lm_off, nullptr, nullptr, nullptr, logits_out, B, c->V, H, EPI_NONE, FVHD_F32), "lm_head gemm");
    return 0;
}

// hidden states after the decoder stack (before the final norm) of the last prefill: [batch * seq_len, hidden] bf16, for tests
int fvhd_llm_debug_hidden(fvhd_llm* c, void* out, int rows, fvhd_stream_t stream)
{
    if (!c || !out || rows <= 0 || rows > c->ws_rows) return lfail("fvhd_llm_debug_hidden: bad argument");
    DevGuard g(c->device);
    hipError_t e = hipMemcpyAsync(out, c->h, (size_t)rows * c->H * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    return e == hipSuccess ? 0 : lhip("hipMemcpyAsync", e);
}

// ---- single ops (unit tests) ----
int fvhd_op_rmsnorm(fvhd_stream_t st, const void* x, void* y, const float* w, int M, int H, float eps)
{
    if (!x || !y || !w) return lfail("fvhd_op_rmsnorm: NULL pointer");
    int e = fvhd_launch_rmsnorm((hipStream_t)st, x, y, w, M, H, eps);
    return e ? lhip("fvhd_op_rmsnorm", (hipError_t)e) : 0;
}

int fvhd_op_rope(fvhd_stream_t st, void* qkv, const int64_t* pos, const float* table, void* k_cache, void* v_cache, int M, int T, int n_heads,
                 int n_kv_heads, int head_dim, int table_positions, float rope_theta)
{
    if (!qkv || !table) return lfail("fvhd_op_rope: NULL pointer");
    int e = fvhd_launch_rope((hipStream_t)st, qkv, (const long*)pos, table, k_cache, v_cache, M, T, n_heads, n_kv_heads, head_dim, table_positions, rope_theta);
    return e ? lhip("fvhd_op_rope", (hipError_t)e) : 0;
}

int fvhd_op_gemm_qkv_rope(fvhd_stream_t st, const void* A, const void* Wt, const float* bias, void* out, int Mp, int N, int K, const int64_t* pos,
                          const float* table, void* k_cache, void* v_cache, int M, int T, int n_heads, int n_kv_heads, int head_dim, int table_positions,
                          float rope_theta)
{
    if (!A || !Wt || !bias || !out || !table) return lfail("fvhd_op_gemm_qkv_rope: NULL pointer");
    if (!fvhd_gemm_qkv_rope_supported(Mp, N, K, head_dim, n_heads, n_kv_heads))
        return lfail("fvhd_op_gemm_qkv_rope: needs head_dim 64, N = (n_heads + 2 n_kv_heads) * 64, Mp % 128 == 0, N % 128 == 0, K % 64 == 0 and at most one "
                     "128 x 128 tile per CU (fvhd_gemm_qkv_rope_supported); other shapes run fvhd_op_gemm + fvhd_op_rope");
    int e = fvhd_launch_gemm_qkv_rope((hipStream_t)st, A, Wt, bias, out, Mp, N, K, (const long*)pos, table, k_cache, v_cache, M, T, n_heads, n_kv_heads,
                                      head_dim, table_positions, rope_theta);
    return e ? lhip("fvhd_op_gemm_qkv_rope", (hipError_t)e) : 0;
}

int fvhd_op_gemm_splitk(fvhd_stream_t st, const void* A, const void* Wt, const void* resid, void* out, float* partial, int M, int N, int K, int splits)
{
    if (!A || !Wt || !out || !partial) return lfail("fvhd_op_gemm_splitk: NULL pointer");
    if (splits < 1 || N % 128 || K % (64 * splits)) return lfail("fvhd_op_gemm_splitk: needs N % 128 == 0 and K % (64 * splits) == 0");
    int e = fvhd_launch_gemm_splitk((hipStream_t)st, A, Wt, resid, out, partial, M, N, K, splits);
    return e ? lhip("fvhd_op_gemm_splitk", (hipError_t)e) : 0;
}

int fvhd_op_gemm_splitk_norm(fvhd_stream_t st, const void* A, const void* Wt, const void* resid, void* out, float* partial, int M, int N, int K, int splits,
                             const float* norm_w, void* norm_out, float eps)
{
    if (!A || !Wt || !out || !partial || !norm_w || !norm_out) return lfail("fvhd_op_gemm_splitk_norm: NULL pointer");
    if (norm_out == out) return lfail("fvhd_op_gemm_splitk_norm: norm_out must not alias out");
    if (splits < 1 || N % 128 || K % (64 * splits)) return lfail("fvhd_op_gemm_splitk_norm: needs N % 128 == 0 and K % (64 * splits) == 0");
    int e = fvhd_launch_gemm_splitk_norm((hipStream_t)st, A, Wt, resid, out, partial, M, N, K, splits, norm_w, norm_out, eps);
    return e ? lhip("fvhd_op_gemm_splitk_norm", (hipError_t)e) : 0;
}

int fvhd_op_qkv_splitk_rope(fvhd_stream_t st, const void* A, const void* Wt, const float* bias, float* partial, void* qkv, const int64_t* pos,
                            const float* table, void* k_cache, void* v_cache, int M, int Mp, int K, int T, int n_heads, int n_kv_heads, int head_dim,
                            int table_positions, float rope_theta, int splits)
{
    if (!A || !Wt || !partial || !qkv || !table) return lfail("fvhd_op_qkv_splitk_rope: NULL pointer");
    const int width = (n_heads + 2 * n_kv_heads) * head_dim;
    if (splits < 1 || width % 128 || K % (64 * splits) || Mp < M) return lfail("fvhd_op_qkv_splitk_rope: needs width % 128 == 0, K % (64 * splits) == 0, Mp >= M");
    int e = fvhd_launch_gemm_splitk_partials((hipStream_t)st, A, Wt, partial, Mp, width, K, splits);
    if (e) return lhip("fvhd_op_qkv_splitk_rope (gemm)", (hipError_t)e);
    e = fvhd_launch_splitk_bias_rope((hipStream_t)st, partial, splits, Mp, bias, qkv, (const long*)pos, table, k_cache, v_cache, M, T, n_heads, n_kv_heads,
                                     head_dim, table_positions, rope_theta);
    return e ? lhip("fvhd_op_qkv_splitk_rope (reduce)", (hipError_t)e) : 0;
}

int fvhd_op_attention_causal(fvhd_stream_t st, const void* qkv, void* out, const uint8_t* key_valid, int B, int T, int n_heads, int n_kv_heads, int head_dim)
{
    if (!qkv || !out) return lfail("fvhd_op_attention_causal: NULL pointer");
    int e = fvhd_launch_llm_attention((hipStream_t)st, qkv, out, key_valid, B, T, n_heads, n_kv_heads, head_dim);
    return e ? lhip("fvhd_op_attention_causal", (hipError_t)e) : 0;
}

}  // extern "C"
