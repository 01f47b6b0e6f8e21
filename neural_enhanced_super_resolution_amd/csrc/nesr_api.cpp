// C ABI of libnesr_hip.so (include/nesr_hip.h): context, strict weight loading + repacking,
// workspace, and the RRDBNet forward graph as a sequence of fused conv launches.
//
// What it stands behind in the reference: basicsr RRDBNet.__init__/forward and realesrgan
// RealESRGANer's load_state_dict, as called from nesr/nesr.py:216-229,887-891 and
// standalone/direct_esrgan.py:104-148.
#include "../../include/nesr_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "nesr_kernels.h"

using namespace nesr;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail(NESR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));   \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

struct Layer {
    std::string name;
    int cin = 0, cout = 0, cin_p = 0, cout_p = 0;
    std::vector<float> w, b;  // host copies until finalize
    bool has_w = false, has_b = false;
    void* d_w = nullptr;
    void* d_ww = nullptr;   // Winograd-transformed weights (f32 Winograd contexts)
    float* d_b = nullptr;
};

}  // namespace

// byte offsets of the feature maps inside the workspace
struct WsLayout {
    size_t in, f, a, b, c, u1, u2, u3, sync, total;
    int sync_words;
};
// geometry + workspace views of one evaluation (see fw_* below)
struct FwState {
    int N = 0, h = 0, w = 0;      // batch, internal (trunk) height and width
    WsLayout L;
    nesr::Map m_in, m_f, m_t, m_u1, m_u2;
    char* buf[3] = {nullptr, nullptr, nullptr};
};

struct nesr_ctx {
    int device = 0, cin0 = 3, unshuffle = 0, nf = 64, nb = 23, gc = 32, nout = 3, dtype = 0;
    bool winograd = false;   // f32 feature-map convs by Winograd F(2x2,3x3) (NESR_DTYPE_F32_WINOGRAD)
    int kgroup = 8;  // K-group of the conv kernel: cin padding granule
    std::vector<Layer> layers;
    std::unordered_map<std::string, int> index;
    bool finalized = false;
    char* d_weights = nullptr;   // arena: [256 B of zeros | packed weights and biases]
    char* ws = nullptr;
    size_t ws_bytes = 0;
    TrunkLayer* d_trunk = nullptr;
    unsigned* last_sync = nullptr;   // abort word of the most recent persistent launch   // device copy of the trunk's layer table (persistent trunk kernel)
    int trunk_mode = 0;              // 0 auto, 1 per-layer launches, 2 persistent kernel
    int shared_device = 0;           // nesr_set_concurrent: other contexts run on the device at the same time
    int size_independent = 0;        // nesr_set_size_independent: kernel choice must not depend on the image size
    // the ragged batch being evaluated (nesr_forward_ragged): internal-resolution sizes of its images
    int rag_n = 0, rag_base_h = 0;
    unsigned short rag_h[nesr::RAG_MAX], rag_w[nesr::RAG_MAX];
    unsigned* d_status = nullptr;    // [0] sticky range word of the f16-pair path (ConvArgs::status), [1] abort word of the fused
                                     // dense-block kernel, [64..] its per-tile progress words
    unsigned rdb_epoch = 0;          // fused dense-block launches: progress values of a launch are epoch+1 .. epoch+4
    int rdb_mode = -1;               // NESR_RDB_FUSE: -1 auto (fuse when every tile gets its own CU), 0 never
    int cus = 256;
    unsigned* h_status = nullptr;    // pinned landing word of nesr_check_range
    // bf16 dense blocks with the working set resident in LDS (rdb_bf16_strip.hip)
    char* d_strip = nullptr;         // per dense block: weight stream (strip_weight_bytes()) + 192 f32 of bias
    size_t strip_stride = 0;
    int strip_mode = -1;             // NESR_STRIP: -1 auto (size-independent contexts, or batches that fill the device), 0 never, 1 wherever it applies
    unsigned strip_epoch = 0;
    bool strip_used = false;         // a strip launch went out since the last status check
    unsigned long long strip_timeout_ticks = 20000000ull;   // 200 ms of s_memrealtime: what an inter-workgroup wait of a persistent kernel may take
    int rdb_mode_init = -1, strip_mode_init = -1;
    int strip_seg = 0;               // NESR_STRIP_SEG: positions per row segment of a strip at most (0: the packer decides, -1: never cut)
    // sharded frames (nesr_comm_init / nesr_forward_sharded_u8): RCCL communicator + scratch
    void* comm = nullptr;            // ncclComm_t
    int comm_rank = 0, comm_nranks = 1;
    char* shard_buf = nullptr;
    size_t shard_bytes = 0;
    int debug_drop = 0;              // nesr_debug_fault: workgroups the next persistent launch leaves out
    int fused_aborts = 0;            // persistent launches that gave up (the context runs per-layer launches from then on)
    struct StripPlan {
        std::vector<int> key;        // N, H, W, then (h, w) of every image
        void* d_items = nullptr; int* d_first = nullptr; char* d_xch = nullptr;
        int grid = 0, smax = 0, makespan = 0;
        double efficiency = 0.0;
    };
    std::vector<StripPlan> strip_plans;
    FwState band;                    // the banded evaluation in progress (nesr_band_*)
    bool band_valid = false;
    // kernel timing hook
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;
    int64_t timed_launches = 0;
    double timed_flops = 0.0;

    size_t esize() const { return dtype == NESR_DTYPE_BF16 ? 2 : 4; }   // bytes per stored activation value
    // activation layout / kernel family: 0 f32 NHWC, 1 bf16 blocked, 2 f16 hi|lo blocked (PackArgs::bf16)
    int kind() const { return dtype == NESR_DTYPE_BF16 ? 1 : (dtype == NESR_DTYPE_F32_SPLIT ? 2 : 0); }
    int ct() const { return nf + 4 * gc; }  // channels of a dense-block buffer
    int ufac() const { return unshuffle > 1 ? unshuffle : 1; }
};

namespace {

int layer_id(const nesr_ctx* c, int b, int r, int k) { return 1 + (b * 3 + r) * 5 + k; }  // r,k zero based

WsLayout ws_layout(const nesr_ctx* c, int N, int h, int w) {
    const size_t es = c->esize();
    const size_t px = (size_t)N * h * w;
    WsLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    L.in = take(px * c->layers[0].cin_p * es);
    L.f = take(px * c->nf * es);
    L.a = take(px * c->ct() * es);
    L.b = take(px * c->ct() * es);
    L.c = take(px * c->ct() * es);
    L.u1 = take(px * 4 * c->nf * es);
    L.u2 = take(px * 16 * c->nf * es);
    L.u3 = take(px * 16 * c->nf * es);
    // per-tile progress counters of the persistent trunk kernel (8x16-pixel tiles) + abort word
    L.sync_words = N * ((h + 7) / 8) * ((w + 15) / 16) + 64;
    L.sync = take((size_t)L.sync_words * 4);
    L.total = off;
    return L;
}

int ensure_ws(nesr_ctx* c, size_t bytes) {
    if (bytes <= c->ws_bytes) return NESR_OK;
    if (c->ws) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(c->ws));
        c->ws = nullptr;
        c->ws_bytes = 0;
        c->last_sync = nullptr;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(NESR_ERR_NOMEM, "hipMalloc(workspace " + std::to_string(bytes) + " B): " + hipGetErrorString(e));
    c->ws = static_cast<char*>(p);
    c->ws_bytes = bytes;
    return NESR_OK;
}

hipError_t launch_conv(const nesr_ctx* c, const ConvArgs& a, hipStream_t s, const Layer* L = nullptr) {
    if (c->dtype == NESR_DTYPE_BF16) return launch_conv3x3_bf16(a, s);
    if (c->dtype == NESR_DTYPE_F32_SPLIT) return launch_conv3x3_f16x2(a, s);
    if (c->winograd && L && L->d_ww && (!(a.out_nchw || a.out_u8) || (a.cout_real >= 1 && a.cout_real <= 4 && a.coutp == 32))) {
        ConvArgs w = a;
        w.w = L->d_ww;
        return launch_conv3x3_wino_f32(w, s);
    }
    return launch_conv3x3_f32(a, s);
}

ConvArgs base_args(const nesr_ctx* c, const Layer& L, int N, int h, int w) {
    ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.zeros = c->d_weights;
    a.cin = L.cin_p;
    a.w = L.d_w;
    a.bias = L.d_b;
    a.coutp = L.cout_p;
    a.n = N;
    a.h = h;
    a.w_ = w;
    a.in_h = h;
    a.in_w = w;
    a.s1 = a.s2 = 1.f;
    a.cout_real = 0;
    a.shared_device = c->shared_device;
    a.size_independent = c->size_independent;
    if (c->rag_n) {
        a.rag_n = c->rag_n;
        a.rag_shift = h == c->rag_base_h ? 0 : (h == 2 * c->rag_base_h ? 1 : 2);
        std::memcpy(a.rag_h, c->rag_h, sizeof(a.rag_h));
        std::memcpy(a.rag_w, c->rag_w, sizeof(a.rag_w));
    }
    a.status = c->dtype == NESR_DTYPE_F32_SPLIT ? c->d_status : nullptr;
    return a;
}

// kind 0 (f32): NHWC (pix = channels of the buffer, chunk = 8).  kind 1 (bf16): channel-blocked
// [C/16][pixels][16].  kind 2 (f16 pairs): [C/16][pixels][16 hi | 16 lo], in 2-byte units.
Map make_map(int kind, int channels, size_t pixels) {
    Map m;
    if (kind == 2) { m.pix = 32; m.chunk = (long long)pixels * 32; }
    else if (kind == 1) { m.pix = 16; m.chunk = (long long)pixels * 16; }
    else { m.pix = channels; m.chunk = 8; }
    return m;
}

double conv_flops(const Layer& L, double pixels) { return 2.0 * 9.0 * L.cin * L.cout * pixels; }

// ---- persistent kernels need the device to themselves: every workgroup of rdb_f16x2_kernel / rdb_bf16_strip_kernel waits for
// other workgroups of the same launch, so two such launches that share the compute units (two contexts on two streams) can
// each hold CUs the other one's missing workgroups need.  Within a process they are therefore serialised per device:
// a stream that is about to launch one first waits for the event recorded behind the previous holder's last launch.
// (Across processes nothing can order them: the kernels bound their waits and raise an abort word, nesr_check_range.)
struct DeviceLease {
    std::mutex mu;
    hipEvent_t ev = nullptr;
    hipStream_t owner = nullptr;
    const nesr_ctx* owner_ctx = nullptr;
    bool pending = false;
};
DeviceLease g_lease[64];

int lease_acquire(const nesr_ctx* c, hipStream_t s) {
    if (c->device < 0 || c->device >= 64) return NESR_OK;
    DeviceLease& L = g_lease[c->device];
    std::lock_guard<std::mutex> lock(L.mu);
    if (L.pending && (L.owner_ctx != c || L.owner != s)) HIP_TRY(hipStreamWaitEvent(s, L.ev, 0));
    return NESR_OK;
}
int lease_release(const nesr_ctx* c, hipStream_t s) {
    if (c->device < 0 || c->device >= 64) return NESR_OK;
    DeviceLease& L = g_lease[c->device];
    std::lock_guard<std::mutex> lock(L.mu);
    if (!L.ev) HIP_TRY(hipEventCreateWithFlags(&L.ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(L.ev, s));
    L.owner = s;
    L.owner_ctx = c;
    L.pending = true;
    return NESR_OK;
}
void lease_forget(const nesr_ctx* c) {
    if (c->device < 0 || c->device >= 64) return;
    DeviceLease& L = g_lease[c->device];
    std::lock_guard<std::mutex> lock(L.mu);
    if (L.owner_ctx == c) { L.owner_ctx = nullptr; L.owner = nullptr; }   // the event stays valid: later holders still wait for it
}

void free_strip_plans(nesr_ctx* c) {
    for (auto& P : c->strip_plans) {
        if (P.d_items) (void)hipFree(P.d_items);
        if (P.d_first) (void)hipFree(P.d_first);
        if (P.d_xch) (void)hipFree(P.d_xch);
    }
    c->strip_plans.clear();
}

// the strip schedule + mailboxes of one batch geometry (cached: a video stream asks for the same one every frame)
int strip_plan_for(nesr_ctx* c, int N, int h, int w, const nesr_ctx::StripPlan** out) {
    std::vector<int> key{N, h, w};
    std::vector<int> hw(2 * (size_t)N);
    for (int i = 0; i < N; ++i) {
        hw[2 * i] = c->rag_n ? c->rag_h[i] : h;
        hw[2 * i + 1] = c->rag_n ? c->rag_w[i] : w;
    }
    key.insert(key.end(), hw.begin(), hw.end());
    for (const auto& P : c->strip_plans)
        if (P.key == key) { *out = &P; return NESR_OK; }
    if (c->strip_plans.size() >= 32) {
        HIP_TRY(hipDeviceSynchronize());
        free_strip_plans(c);
    }
    nesr_ctx::StripPlan P;
    P.key = key;
    const StripSchedule S = strip_schedule(N, hw.data(), c->cus, c->strip_seg);
    P.makespan = S.makespan;
    if (S.makespan > 0 && S.makespan < 250) {       // tags hold position * 8 + layer below 2048
        P.grid = S.grid; P.smax = S.smax; P.efficiency = S.efficiency;
        const size_t xb = (size_t)S.nvimg * S.smax * STRIP_XCH_BYTES;
        HIP_TRY(hipMalloc(&P.d_items, S.items.size() * 4));
        HIP_TRY(hipMalloc((void**)&P.d_first, S.wg_first.size() * 4));
        HIP_TRY(hipMalloc((void**)&P.d_xch, xb));
        HIP_TRY(hipMemcpy(P.d_items, S.items.data(), S.items.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(P.d_first, S.wg_first.data(), S.wg_first.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(P.d_xch, 0, xb));
    } else {
        P.makespan = -1;
    }
    if (getenv("NESR_STRIP_DEBUG"))
        fprintf(stderr, "[nesr] strip plan: %d images (slot %dx%d) as %d row segments: grid %d workgroups, makespan %d positions, efficiency %.3f\n", N, h, w,
                S.nvimg, P.grid, P.makespan, P.efficiency);
    c->strip_plans.push_back(std::move(P));
    *out = &c->strip_plans.back();
    return NESR_OK;
}

// does this evaluation's trunk run as persistent (lease-holding) launches?
bool strip_wanted(const nesr_ctx* c) {
    return c->dtype == NESR_DTYPE_BF16 && c->d_strip && c->strip_mode != 0 && c->nf == 64 && c->gc == 32;
}

// ---- the forward graph in stages (whole-frame forward = all of them in order; the banded multi-GPU mode
// runs them one at a time with a row exchange in between)
int fw_setup(nesr_ctx* c, int N, int C, int H, int W, FwState& F) {
    if (!c->finalized) return fail(NESR_ERR_STATE, "weights not finalized (call nesr_finalize_weights)");
    const int u = c->ufac();
    if (N <= 0 || H <= 0 || W <= 0) return fail(NESR_ERR_ARG, "empty input");
    if (C * u * u != c->cin0)
        return fail(NESR_ERR_ARG, "input has " + std::to_string(C) + " channels; conv_first expects " +
                                      std::to_string(c->cin0) + " after unshuffle " + std::to_string(u));
    if (H % u || W % u) return fail(NESR_ERR_ARG, "H and W must be multiples of the unshuffle factor");
    HIP_TRY(hipSetDevice(c->device));
    F.N = N; F.h = H / u; F.w = W / u;
    F.L = ws_layout(c, N, F.h, F.w);
    int rc = ensure_ws(c, F.L.total);
    if (rc) return rc;
    const int kind = c->kind();
    const size_t P1 = (size_t)N * F.h * F.w;
    F.m_in = make_map(kind, c->layers[0].cin_p, P1);
    F.m_f = make_map(kind, c->nf, P1);
    F.m_t = make_map(kind, c->ct(), P1);
    F.m_u1 = make_map(kind, c->nf, P1 * 4);
    F.m_u2 = make_map(kind, c->nf, P1 * 16);
    F.buf[0] = c->ws + F.L.a; F.buf[1] = c->ws + F.L.b; F.buf[2] = c->ws + F.L.c;
    return NESR_OK;
}

// pack (pixel_unshuffle, layout, u8 normalisation) + conv_first: IN -> P.x0 and F (feat is needed again after the trunk)
int fw_first(nesr_ctx* c, const FwState& F, const float* x_f32, const uint8_t* x_u8, int flip, int C, int H, int W, hipStream_t s) {
    char* ws = c->ws;
    PackArgs p;
    std::memset(&p, 0, sizeof(p));
    p.src = x_u8 ? static_cast<const void*>(x_u8) : static_cast<const void*>(x_f32);
    p.src_u8 = x_u8 ? 1 : 0;
    p.flip = flip;
    p.n = F.N; p.c = C; p.hin = H; p.win = W;
    p.unshuffle = c->ufac();
    p.dst = ws + F.L.in;
    p.dst_map = F.m_in;
    p.cp = c->layers[0].cin_p;
    p.bf16 = c->kind();
    p.status = c->dtype == NESR_DTYPE_F32_SPLIT ? c->d_status : nullptr;
    if (p.status) HIP_TRY(launch_status_latch(c->d_status, s));      // the range word is per forward (nesr_check_range reports a latched one once)
    HIP_TRY(launch_pack_input(p, s));
    ConvArgs a = base_args(c, c->layers[0], F.N, F.h, F.w);
    a.in = ws + F.L.in; a.in_map = F.m_in;
    a.out = ws + F.L.a; a.out_map = F.m_t; a.out_coff = 0;
    a.out2 = ws + F.L.f; a.out2_map = F.m_f;
    HIP_TRY(launch_conv(c, a, s, &c->layers[0]));
    return NESR_OK;
}

// RDB r (0..2) of RRDB b.  Buffers P,Q,R hold x0|x1|x2|x3|x4 of RDB1,2,3; RDB3's conv5 applies both residuals
// (x5*0.2+x0 then *0.2 + RRDB input) and lands in P.x0 in place, so every RRDB starts and ends in P.
// phase: -1 the whole block; 0 conv1..conv4 and conv5 on the `edge` band rows next to each apron (what the neighbours
// wait for); 1 conv5 on the band rows in between.  Phases need the f16-pair kernel's row ranges: for the other dtypes
// phase 0 is the whole block and phase 1 nothing.  `top` / `bottom` = apron rows of the band image (conv5 skips them in
// the phased form: they are overwritten by the neighbours' rows before anything reads them).
int fw_rdb(nesr_ctx* c, const FwState& F, int b, int r, hipStream_t s, int phase = -1, int top = 0, int bottom = 0, int edge = 0) {
    const int nf = c->nf, gc = c->gc;
    const double px = (double)F.N * F.h * F.w;
    char* cur = F.buf[r];
    const bool ranged = phase >= 0 && c->dtype == NESR_DTYPE_F32_SPLIT && F.N == 1 && F.h >= top + bottom + 2 * edge;
    if (phase == 1 && !ranged) return NESR_OK;
    if (phase >= 0 && !ranged) phase = -1;
    // bf16: the dense block with its working set resident in LDS (rdb_bf16_strip_kernel), whenever the context is
    // size-independent (a tiling wrapper: one arithmetic for every tile, however it is batched) or the batch fills the device
    if (phase < 0 && strip_wanted(c)) {
        const nesr_ctx::StripPlan* P = nullptr;
        int rc = strip_plan_for(c, F.N, F.h, F.w, &P);
        if (rc) return rc;
        if (P->makespan > 0 && (c->strip_mode == 1 || c->size_independent || P->efficiency >= 0.55)) {
            StripLaunch L;
            std::memset(&L, 0, sizeof(L));
            L.cur = cur;
            L.chunk_bytes = F.m_t.chunk * 2;
            L.out = r < 2 ? F.buf[r + 1] : F.buf[0];
            L.res2 = r < 2 ? nullptr : F.buf[0];
            L.s1 = 0.2f; L.s2 = 0.2f;
            const char* blk = c->d_strip + (size_t)(b * 3 + r) * c->strip_stride;
            L.wstream = blk;
            L.bias = reinterpret_cast<const float*>(blk + strip_weight_bytes());
            L.H = F.h; L.W = F.w;
            L.items = P->d_items; L.wg_first = P->d_first; L.grid = P->grid; L.smax = P->smax; L.xch = P->d_xch;
            c->strip_epoch += 2048;
            L.epoch = c->strip_epoch;
            L.abort_flag = c->d_status + 2;
            L.timeout_ticks = c->strip_timeout_ticks;
            L.debug_drop = c->debug_drop;
            c->debug_drop = 0;
            const hipError_t le = launch_rdb_bf16_strip(L, s);
            if (le == hipErrorLaunchOutOfResources) {
                c->strip_mode = 0;      // the device does not admit the kernel's workgroups (LDS / registers): per-layer launches
                return fw_rdb(c, F, b, r, s, phase, top, bottom, edge);
            }
            HIP_TRY(le);
            c->strip_used = true;
            if (c->timing) {
                double px_real = 0.0;      // ragged batches: the images' own pixels
                for (int i = 0; i < F.N; ++i) px_real += c->rag_n ? (double)c->rag_h[i] * c->rag_w[i] : (double)F.h * F.w;
                for (int k = 0; k < 5; ++k) c->timed_flops += conv_flops(c->layers[layer_id(c, b, r, k)], px_real);
                c->timed_launches += 1;
            }
            return NESR_OK;
        }
    }
    // small frames, f16-pair form: the whole dense block in one launch (rdb_f16x2_kernel).  Every tile needs its own
    // resident workgroup, so the frame's tiles must fit the compute units and the device must be this context's
    // (frames in flight on other streams would compete for the one workgroup slot per CU).
    if (phase < 0 && c->dtype == NESR_DTYPE_F32_SPLIT && c->rdb_mode != 0 && nf == 64 && gc == 32 && !c->shared_device) {
        const int tiles = rdb_f16x2_tiles(F.N, F.h, F.w);
        if (tiles <= c->cus && tiles <= 4096) {
            RdbLaunch L;
            std::memset(&L, 0, sizeof(L));
            L.cur = cur;
            L.chunk_bytes = F.m_t.chunk * 2;
            L.out = r < 2 ? F.buf[r + 1] : F.buf[0];
            L.res2 = r < 2 ? nullptr : F.buf[0];
            L.s1 = 0.2f; L.s2 = 0.2f;
            for (int k = 0; k < 5; ++k) {
                const Layer& Ly = c->layers[layer_id(c, b, r, k)];
                L.w[k] = Ly.d_w;
                L.bias[k] = Ly.d_b;
                if (c->timing) c->timed_flops += conv_flops(Ly, px);
            }
            L.n = F.N; L.h = F.h; L.w_ = F.w;
            L.progress = c->d_status + 64;
            c->rdb_epoch += 8;
            L.epoch = c->rdb_epoch;
            L.abort_flag = c->d_status + 1;
            L.status = c->d_status;
            L.timeout_ticks = c->strip_timeout_ticks;
            L.debug_drop = c->debug_drop;
            c->debug_drop = 0;
            const hipError_t le = launch_rdb_f16x2(L, s);
            if (le == hipErrorLaunchOutOfResources) {
                c->rdb_mode = 0;        // fewer resident workgroups than tiles: per-layer launches (the same bits)
                return fw_rdb(c, F, b, r, s, phase, top, bottom, edge);
            }
            HIP_TRY(le);
            c->strip_used = true;       // (the abort word of either persistent kernel is looked at by nesr_check_range)
            if (c->timing) c->timed_launches += 1;
            return NESR_OK;
        }
    }
    for (int k = 0; k < 4 && phase != 1; ++k) {
        const Layer& Ly = c->layers[layer_id(c, b, r, k)];
        ConvArgs a = base_args(c, Ly, F.N, F.h, F.w);
        a.in = cur; a.in_map = F.m_t;
        a.out = cur; a.out_map = F.m_t; a.out_coff = nf + k * gc;
        a.lrelu = 1;
        HIP_TRY(launch_conv(c, a, s, &Ly));
        if (c->timing) c->timed_flops += conv_flops(Ly, px);
    }
    const Layer& L5 = c->layers[layer_id(c, b, r, 4)];
    ConvArgs a = base_args(c, L5, F.N, F.h, F.w);
    a.in = cur; a.in_map = F.m_t;
    a.res1 = cur; a.res1_map = F.m_t; a.s1 = 0.2f;
    if (r < 2) {
        a.out = F.buf[r + 1];
    } else {
        a.out = F.buf[0];
        a.res2 = F.buf[0]; a.res2_map = F.m_t; a.s2 = 0.2f;
    }
    a.out_map = F.m_t; a.out_coff = 0;
    if (phase < 0) {
        HIP_TRY(launch_conv(c, a, s, &L5));
    } else {
        // band rows [top, h - bottom); a side without an apron (a frame edge) has no neighbour waiting: its rows belong
        // to the interior launch
        const int lo = top, hi = F.h - bottom;
        const int e0 = top ? lo + edge : lo, e1 = bottom ? hi - edge : hi;      // interior = [e0, e1)
        auto rows = [&](int y0, int y1) -> int {
            if (y1 <= y0) return NESR_OK;
            ConvArgs q = a;
            q.y_lo = y0; q.y_hi = y1;
            HIP_TRY(launch_conv(c, q, s, &L5));
            return NESR_OK;
        };
        int rc;
        if (phase == 0) {
            if (top && (rc = rows(lo, e0 < e1 ? e0 : e1))) return rc;
            if (bottom && (rc = rows(e1 > e0 ? e1 : e0, hi))) return rc;
        } else if ((rc = rows(e0, e1))) {
            return rc;
        }
    }
    if (c->timing && phase != 0) { c->timed_flops += conv_flops(L5, px); c->timed_launches += 5; }
    return NESR_OK;
}

// conv_body + trunk skip, the two nearest-x2 + conv stages, conv_hr, conv_last
int fw_tail(nesr_ctx* c, const FwState& F, float* y_f32, uint8_t* y_u8, int flip, int round_mode, hipStream_t s) {
    char* ws = c->ws;
    const int N = F.N, h = F.h, w = F.w;
    const int tail = 1 + c->nb * 15;
    {   // feat = feat + conv_body(trunk)   (in place on F)
        ConvArgs a = base_args(c, c->layers[tail], N, h, w);
        a.in = F.buf[0]; a.in_map = F.m_t;
        a.out = ws + F.L.f; a.out_map = F.m_f;
        a.res1 = ws + F.L.f; a.res1_map = F.m_f; a.s1 = 1.0f;
        HIP_TRY(launch_conv(c, a, s, &c->layers[tail]));
    }
    {   // lrelu(conv_up1(nearest2x(feat)))
        ConvArgs a = base_args(c, c->layers[tail + 1], N, 2 * h, 2 * w);
        a.in = ws + F.L.f; a.in_map = F.m_f; a.in_h = h; a.in_w = w; a.up = 1;
        a.out = ws + F.L.u1; a.out_map = F.m_u1; a.lrelu = 1;
        HIP_TRY(launch_conv(c, a, s, &c->layers[tail + 1]));
    }
    {   // lrelu(conv_up2(nearest2x(feat)))
        ConvArgs a = base_args(c, c->layers[tail + 2], N, 4 * h, 4 * w);
        a.in = ws + F.L.u1; a.in_map = F.m_u1; a.in_h = 2 * h; a.in_w = 2 * w; a.up = 1;
        a.out = ws + F.L.u2; a.out_map = F.m_u2; a.lrelu = 1;
        HIP_TRY(launch_conv(c, a, s, &c->layers[tail + 2]));
    }
    {   // lrelu(conv_hr(feat))
        ConvArgs a = base_args(c, c->layers[tail + 3], N, 4 * h, 4 * w);
        a.in = ws + F.L.u2; a.in_map = F.m_u2;
        a.out = ws + F.L.u3; a.out_map = F.m_u2; a.lrelu = 1;
        HIP_TRY(launch_conv(c, a, s, &c->layers[tail + 3]));
    }
    {   // conv_last -> planar f32 NCHW, or clamped + quantised u8 HWC
        ConvArgs a = base_args(c, c->layers[tail + 4], N, 4 * h, 4 * w);
        a.in = ws + F.L.u3; a.in_map = F.m_u2;
        a.cout_real = c->nout;
        a.out_nchw = y_f32;
        a.out_u8 = y_u8;
        a.u8_flip = flip;
        a.u8_round = round_mode;
        HIP_TRY(launch_conv(c, a, s, &c->layers[tail + 4]));
    }
    return NESR_OK;
}

// The whole forward.  x -> y; exactly one of (x_f32, x_u8) and one of (y_f32, y_u8) is set.
int run_forward(nesr_ctx* c, const float* x_f32, const uint8_t* x_u8, int flip, int N, int C, int H, int W,
                float* y_f32, uint8_t* y_u8, int round_mode, hipStream_t s) {
    FwState F;
    int rc = fw_setup(c, N, C, H, W, F);
    if (rc) return rc;
    c->band_valid = false;   // the workspace no longer holds a banded evaluation
    if ((rc = fw_first(c, F, x_f32, x_u8, flip, C, H, W, s))) return rc;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (c->timing) {
        if (!c->ev_free.empty()) {
            ev0 = c->ev_free.back().first; ev1 = c->ev_free.back().second;
            c->ev_free.pop_back();
        } else {
            HIP_TRY(hipEventCreate(&ev0));
            HIP_TRY(hipEventCreate(&ev1));
        }
        HIP_TRY(hipEventRecord(ev0, s));
    }
    const bool persist = c->trunk_mode == 2;   // opt-in (NESR_TRUNK=persist): measured slower at 2 tiles/CU, see DESIGN.md
    if (persist && c->nb > 0) {
        // one cooperative launch for all 15*nb dense-block convs (tile-level dataflow sync)
        unsigned* sync = reinterpret_cast<unsigned*>(c->ws + F.L.sync);
        HIP_TRY(hipMemsetAsync(sync, 0, (size_t)F.L.sync_words * 4, s));
        TrunkArgs t;
        std::memset(&t, 0, sizeof(t));
        t.layers = c->d_trunk;
        t.nlayers = c->nb * 15;
        t.buf[0] = F.buf[0]; t.buf[1] = F.buf[1]; t.buf[2] = F.buf[2];
        t.map = F.m_t;
        t.n = N; t.h = F.h; t.w = F.w;
        t.progress = sync + 64;
        t.abort_flag = sync;
        t.zeros = c->d_weights;
        HIP_TRY(launch_trunk_persist(t, c->kind() == 1, s));
        c->last_sync = sync;
        if (c->timing) {
            for (int i = 0; i < c->nb * 15; ++i) c->timed_flops += conv_flops(c->layers[1 + i], (double)N * F.h * F.w);
            c->timed_launches += 1;
        }
    } else {
        // the fused dense-block kernels hold the device: serialised per device against other streams' (see DeviceLease)
        const bool lease = (strip_wanted(c) || (c->dtype == NESR_DTYPE_F32_SPLIT && c->rdb_mode != 0 && !c->shared_device)) && c->nb > 0;
        if (lease && (rc = lease_acquire(c, s))) return rc;
        for (int b = 0; b < c->nb; ++b)
            for (int r = 0; r < 3; ++r)
                if ((rc = fw_rdb(c, F, b, r, s))) return rc;
        if (lease && (rc = lease_release(c, s))) return rc;
    }
    if (c->timing) {
        HIP_TRY(hipEventRecord(ev1, s));
        c->ev_pending.emplace_back(ev0, ev1);
    }
    return fw_tail(c, F, y_f32, y_u8, flip, round_mode, s);
}

}  // namespace

extern "C" {

const char* nesr_last_error(void) { return g_err.c_str(); }
const char* nesr_version(void) { return "nesr_hip 0.1 (gfx950)"; }

int nesr_create(nesr_ctx** out, int device_id, int conv_first_in_ch, int unshuffle, int num_feat, int num_block,
                int num_grow_ch, int num_out_ch, int dtype) {
    if (!out) return fail(NESR_ERR_ARG, "out is null");
    *out = nullptr;
    if (unshuffle != 0 && unshuffle != 1 && unshuffle != 2 && unshuffle != 4)
        return fail(NESR_ERR_ARG, "unshuffle must be 0, 2 or 4");
    const int u = unshuffle > 1 ? unshuffle : 1;
    if (conv_first_in_ch <= 0 || conv_first_in_ch % (u * u))
        return fail(NESR_ERR_ARG, "conv_first_in_ch must be a positive multiple of unshuffle^2");
    if (num_feat != 32 && num_feat != 64) return fail(NESR_ERR_ARG, "num_feat must be 32 or 64 (reference uses 64)");
    if (num_grow_ch != 32) return fail(NESR_ERR_ARG, "num_grow_ch must be 32 (reference uses 32)");
    if (num_block < 0 || num_out_ch <= 0 || num_out_ch > 32) return fail(NESR_ERR_ARG, "bad num_block / num_out_ch");
    if (dtype != NESR_DTYPE_F32 && dtype != NESR_DTYPE_BF16 && dtype != NESR_DTYPE_F32_WINOGRAD && dtype != NESR_DTYPE_F32_SPLIT)
        return fail(NESR_ERR_ARG, "dtype must be 0 (f32 direct), 1 (bf16), 2 (f32 Winograd) or 3 (f32 as f16 pairs)");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(NESR_ERR_ARG, "no such device " + std::to_string(device_id));
    HIP_TRY(hipSetDevice(device_id));
    unsigned* d_status = nullptr;
    unsigned* h_status = nullptr;
    constexpr size_t STATUS_BYTES = 256 + 4096 * 4;   // status words + progress words of up to 4096 tiles
    HIP_TRY(hipMalloc((void**)&d_status, STATUS_BYTES));
    if (hipMemset(d_status, 0, STATUS_BYTES) != hipSuccess || hipHostMalloc((void**)&h_status, 64, hipHostMallocDefault) != hipSuccess) {
        (void)hipFree(d_status);
        return fail(NESR_ERR_HIP, "allocating the context's status words failed");
    }
    *h_status = 0;
    nesr_ctx* c = new nesr_ctx();
    c->d_status = d_status;
    c->h_status = h_status;
    c->device = device_id;
    c->cin0 = conv_first_in_ch;
    c->unshuffle = unshuffle;
    c->nf = num_feat;
    c->nb = num_block;
    c->gc = num_grow_ch;
    c->nout = num_out_ch;
    c->winograd = dtype == NESR_DTYPE_F32_WINOGRAD;
    if (const char* e = getenv("NESR_F32_ALGO")) {   // override for A/B timing: direct | winograd
        if (dtype != NESR_DTYPE_BF16) {
            c->winograd = e[0] == 'w';
            dtype = e[0] == 's' ? NESR_DTYPE_F32_SPLIT : (e[0] == 'w' ? NESR_DTYPE_F32_WINOGRAD : NESR_DTYPE_F32);
        }
    }
    if (dtype == NESR_DTYPE_F32_WINOGRAD) dtype = NESR_DTYPE_F32;
    c->dtype = dtype;
    c->kgroup = (dtype == NESR_DTYPE_BF16 || dtype == NESR_DTYPE_F32_SPLIT) ? 16 : 8;
    if (const char* e = getenv("NESR_TRUNK")) c->trunk_mode = e[0] == 'l' ? 1 : (e[0] == 'p' ? 2 : 0);
    if (const char* e = getenv("NESR_RDB_FUSE")) c->rdb_mode = atoi(e);
    if (const char* e = getenv("NESR_STRIP")) c->strip_mode = atoi(e);
    c->rdb_mode_init = c->rdb_mode;
    c->strip_mode_init = c->strip_mode;
    if (const char* e = getenv("NESR_STRIP_SEG")) c->strip_seg = atoi(e);
    if (const char* e = getenv("NESR_FUSED_TIMEOUT_MS")) c->strip_timeout_ticks = (unsigned long long)atoll(e) * 100000ull;
    (void)hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, device_id);
    auto add = [&](const std::string& name, int cin, int cout) {
        Layer L;
        L.name = name;
        L.cin = cin;
        L.cout = cout;
        L.cin_p = round_up(cin, c->kgroup);
        L.cout_p = round_up(cout, 32);
        c->index[name] = (int)c->layers.size();
        c->layers.push_back(std::move(L));
    };
    add("conv_first", c->cin0, c->nf);
    for (int b = 0; b < c->nb; ++b)
        for (int r = 1; r <= 3; ++r) {
            const std::string pre = "body." + std::to_string(b) + ".rdb" + std::to_string(r) + ".conv";
            for (int k = 1; k <= 4; ++k) add(pre + std::to_string(k), c->nf + (k - 1) * c->gc, c->gc);
            add(pre + "5", c->nf + 4 * c->gc, c->nf);
        }
    add("conv_body", c->nf, c->nf);
    add("conv_up1", c->nf, c->nf);
    add("conv_up2", c->nf, c->nf);
    add("conv_hr", c->nf, c->nf);
    add("conv_last", c->nf, c->nout);
    *out = c;
    return NESR_OK;
}

int nesr_num_tensors(const nesr_ctx* c) { return c ? (int)c->layers.size() * 2 : 0; }

int nesr_load_weight(nesr_ctx* c, const char* key, const float* data, const int64_t* shape, int ndim) {
    if (!c || !key || !data || !shape) return fail(NESR_ERR_ARG, "null argument");
    std::string k(key);
    const size_t dot = k.rfind('.');
    if (dot == std::string::npos) return fail(NESR_ERR_ARG, "unexpected key in state_dict: " + k);
    const std::string lname = k.substr(0, dot), kind = k.substr(dot + 1);
    auto it = c->index.find(lname);
    if (it == c->index.end() || (kind != "weight" && kind != "bias"))
        return fail(NESR_ERR_ARG, "unexpected key in state_dict: " + k);
    Layer& L = c->layers[it->second];
    if (kind == "weight") {
        if (ndim != 4 || shape[0] != L.cout || shape[1] != L.cin || shape[2] != 3 || shape[3] != 3)
            return fail(NESR_ERR_ARG, "size mismatch for " + k + ": expected [" + std::to_string(L.cout) + "," +
                                          std::to_string(L.cin) + ",3,3]");
        L.w.assign(data, data + (size_t)L.cout * L.cin * 9);
        L.has_w = true;
    } else {
        if (ndim != 1 || shape[0] != L.cout)
            return fail(NESR_ERR_ARG, "size mismatch for " + k + ": expected [" + std::to_string(L.cout) + "]");
        L.b.assign(data, data + L.cout);
        L.has_b = true;
    }
    c->finalized = false;
    return NESR_OK;
}

int nesr_finalize_weights(nesr_ctx* c) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    std::string missing;
    int nmiss = 0;
    for (const Layer& L : c->layers) {
        if (!L.has_w && nmiss++ < 4) missing += " " + L.name + ".weight";
        if (!L.has_b && nmiss++ < 4) missing += " " + L.name + ".bias";
    }
    if (nmiss) return fail(NESR_ERR_STATE, "Missing key(s) in state_dict (" + std::to_string(nmiss) + "):" + missing);
    // non-finite parameters are refused for every dtype; the f16-pair form also needs |w| <= 65504 (the hi half
    // is an f16) -- never a silently clamped weight
    for (const Layer& L : c->layers) {
        const float lim = c->dtype == NESR_DTYPE_F32_SPLIT ? 65504.f : INFINITY;
        for (int t = 0; t < 2; ++t) {
            const std::vector<float>& v = t ? L.b : L.w;
            const float blim = t ? INFINITY : lim;   // biases are added in f32
            for (size_t i = 0; i < v.size(); ++i)
                if (!(std::fabs(v[i]) <= blim) || !std::isfinite(v[i]))
                    return fail(NESR_ERR_RANGE, L.name + (t ? ".bias" : ".weight") + "[" + std::to_string(i) + "] = " + std::to_string(v[i]) +
                                                    (std::isfinite(v[i]) ? ": |w| > 65504 does not fit the f16-pair form of compute_dtype f32 "
                                                                           "(use f32-winograd or f32-direct)" : ": non-finite parameter"));
        }
    }
    HIP_TRY(hipSetDevice(c->device));
    const bool bf = c->dtype == NESR_DTYPE_BF16;
    const bool sp = c->dtype == NESR_DTYPE_F32_SPLIT;
    size_t total = 256;   // leading zero page
    std::vector<size_t> woff(c->layers.size()), boff(c->layers.size()), wwoff(c->layers.size(), 0);
    const size_t last = c->layers.size() - 1;
    for (size_t i = 0; i < c->layers.size(); ++i) {
        const Layer& L = c->layers[i];
        const size_t we = sp ? packed_weight_elems_f16x2(L.cin_p, L.cout_p)
                             : (bf ? packed_weight_elems_bf16(L.cin_p, L.cout_p) : packed_weight_elems_f32(L.cin_p, L.cout_p));
        woff[i] = total;
        total = align_up(total + we * ((bf || sp) ? 2 : 4), 256);
        boff[i] = total;
        total = align_up(total + (size_t)L.cout_p * 4, 256);
        (void)last;
        if (c->winograd) {
            wwoff[i] = total;
            total = align_up(total + packed_weight_elems_wino_f32(L.cin_p, L.cout_p) * 4, 256);
        }
    }
    std::vector<char> host(total, 0);
    for (size_t i = 0; i < c->layers.size(); ++i) {
        const Layer& L = c->layers[i];
        if (sp)
            pack_weights_f16x2(L.w.data(), L.cout, L.cin, L.cin_p, L.cout_p, reinterpret_cast<uint16_t*>(host.data() + woff[i]));
        else if (bf)
            pack_weights_bf16(L.w.data(), L.cout, L.cin, L.cin_p, L.cout_p, reinterpret_cast<uint16_t*>(host.data() + woff[i]));
        else
            pack_weights_f32(L.w.data(), L.cout, L.cin, L.cin_p, L.cout_p, reinterpret_cast<float*>(host.data() + woff[i]));
        std::memcpy(host.data() + boff[i], L.b.data(), (size_t)L.cout * 4);
        if (wwoff[i]) pack_weights_wino_f32(L.w.data(), L.cout, L.cin, L.cin_p, L.cout_p, reinterpret_cast<float*>(host.data() + wwoff[i]));
    }
    if (c->d_weights) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(c->d_weights));
        c->d_weights = nullptr;
    }
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, total));
    c->d_weights = static_cast<char*>(p);
    HIP_TRY(hipMemcpy(c->d_weights, host.data(), total, hipMemcpyHostToDevice));
    for (size_t i = 0; i < c->layers.size(); ++i) {
        c->layers[i].d_w = c->d_weights + woff[i];
        c->layers[i].d_b = reinterpret_cast<float*>(c->d_weights + boff[i]);
        c->layers[i].d_ww = wwoff[i] ? c->d_weights + wwoff[i] : nullptr;
    }
    // bf16: every dense block's weights once more as the LDS-resident kernel's stream (rdb_bf16_strip.hip), + its 192 biases
    if (c->d_strip) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->d_strip)); c->d_strip = nullptr; }
    if (bf && c->nf == 64 && c->gc == 32 && c->nb > 0) {
        c->strip_stride = align_up(strip_weight_bytes() + 192 * 4, 256);
        std::vector<char> hs((size_t)c->nb * 3 * c->strip_stride, 0);
        for (int b = 0; b < c->nb; ++b)
            for (int r = 0; r < 3; ++r) {
                char* blk = hs.data() + (size_t)(b * 3 + r) * c->strip_stride;
                const float* w5[5];
                float* bias = reinterpret_cast<float*>(blk + strip_weight_bytes());
                for (int k = 0; k < 5; ++k) {
                    const Layer& Ly = c->layers[layer_id(c, b, r, k)];
                    w5[k] = Ly.w.data();
                    std::memcpy(bias + 32 * k, Ly.b.data(), (size_t)Ly.cout * 4);
                }
                pack_strip_weights(w5, reinterpret_cast<uint16_t*>(blk));
            }
        HIP_TRY(hipMalloc((void**)&c->d_strip, hs.size()));
        HIP_TRY(hipMemcpy(c->d_strip, hs.data(), hs.size(), hipMemcpyHostToDevice));
    }
    // layer table of the persistent trunk kernel (same wiring as the per-layer loop in run_forward)
    {
        std::vector<TrunkLayer> tl;
        for (int b = 0; b < c->nb; ++b)
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 5; ++k) {
                    const Layer& Ly = c->layers[layer_id(c, b, r, k)];
                    TrunkLayer t;
                    std::memset(&t, 0, sizeof(t));
                    t.in_buf = r;
                    t.cin = Ly.cin_p;
                    t.coutp = Ly.cout_p;
                    t.w = Ly.d_w;
                    t.bias = Ly.d_b;
                    t.res1_buf = t.res2_buf = -1;
                    t.s1 = t.s2 = 1.f;
                    if (k < 4) {
                        t.out_buf = r; t.out_coff = c->nf + k * c->gc; t.lrelu = 1;
                    } else {
                        t.out_coff = 0; t.lrelu = 0;
                        t.res1_buf = r; t.s1 = 0.2f;
                        if (r < 2) {
                            t.out_buf = r + 1;
                        } else {
                            t.out_buf = 0; t.res2_buf = 0; t.s2 = 0.2f;
                        }
                    }
                    tl.push_back(t);
                }
        if (c->d_trunk) { HIP_TRY(hipFree(c->d_trunk)); c->d_trunk = nullptr; }
        if (!tl.empty()) {
            HIP_TRY(hipMalloc((void**)&c->d_trunk, tl.size() * sizeof(TrunkLayer)));
            HIP_TRY(hipMemcpy(c->d_trunk, tl.data(), tl.size() * sizeof(TrunkLayer), hipMemcpyHostToDevice));
        }
    }
    c->finalized = true;
    return NESR_OK;
}

int nesr_forward(nesr_ctx* c, const void* x_dev, int N, int C, int H, int W, void* y_dev, void* stream) {
    if (!c || !x_dev || !y_dev) return fail(NESR_ERR_ARG, "null argument");
    return run_forward(c, static_cast<const float*>(x_dev), nullptr, 0, N, C, H, W, static_cast<float*>(y_dev), nullptr, 0,
                       static_cast<hipStream_t>(stream));
}

int nesr_forward_ragged(nesr_ctx* c, const void* x_dev, int N, int C, int H, int W, const int* hw, void* y_dev, void* stream) {
    if (!c || !x_dev || !y_dev || !hw) return fail(NESR_ERR_ARG, "null argument");
    if (c->dtype != NESR_DTYPE_BF16) return fail(NESR_ERR_ARG, "nesr_forward_ragged: compute dtype bf16 only (the other forms batch equal-sized images)");
    if (N < 1 || N > nesr::RAG_MAX) return fail(NESR_ERR_ARG, "nesr_forward_ragged: 1.." + std::to_string(nesr::RAG_MAX) + " images per call");
    const int u = c->ufac();
    if (H % u || W % u || H / u > 16383 || W / u > 16383) return fail(NESR_ERR_ARG, "nesr_forward_ragged: slot size");
    for (int i = 0; i < N; ++i) {
        const int h = hw[2 * i], w = hw[2 * i + 1];
        if (h < 1 || w < 1 || h > H || w > W || h % u || w % u)
            return fail(NESR_ERR_ARG, "nesr_forward_ragged: image " + std::to_string(i) + " is " + std::to_string(h) + "x" + std::to_string(w) +
                                          ", slot " + std::to_string(H) + "x" + std::to_string(W) + ", unshuffle " + std::to_string(u));
        c->rag_h[i] = (unsigned short)(h / u);
        c->rag_w[i] = (unsigned short)(w / u);
    }
    c->rag_n = N;
    c->rag_base_h = H / u;
    const int rc = run_forward(c, static_cast<const float*>(x_dev), nullptr, 0, N, C, H, W, static_cast<float*>(y_dev), nullptr, 0,
                               static_cast<hipStream_t>(stream));
    c->rag_n = 0;
    return rc;
}

int nesr_set_size_independent(nesr_ctx* c, int on) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    c->size_independent = on ? 1 : 0;
    return NESR_OK;
}

int nesr_forward_u8(nesr_ctx* c, const uint8_t* in_hwc_dev, int H, int W, uint8_t* out_hwc_dev, int flip_rgb,
                    int round_mode, void* stream) {
    if (!c || !in_hwc_dev || !out_hwc_dev) return fail(NESR_ERR_ARG, "null argument");
    const int u = c->ufac();
    if (c->cin0 != 3 * u * u || c->nout != 3)
        return fail(NESR_ERR_ARG, "nesr_forward_u8 needs a 3-channel-in / 3-channel-out network");
    return run_forward(c, nullptr, in_hwc_dev, flip_rgb ? 1 : 0, 1, 3, H, W, nullptr, out_hwc_dev,
                       round_mode == NESR_ROUND_NEAREST ? 1 : 0, static_cast<hipStream_t>(stream));
}

size_t nesr_workspace_bytes(const nesr_ctx* c, int N, int H, int W) {
    if (!c || N <= 0 || H <= 0 || W <= 0) return 0;
    const int u = c->ufac();
    return ws_layout(c, N, (H + u - 1) / u, (W + u - 1) / u).total;
}

int nesr_reserve(nesr_ctx* c, int N, int H, int W) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    return ensure_ws(c, nesr_workspace_bytes(c, N, H, W));
}

double nesr_forward_flops(const nesr_ctx* c, int N, int H, int W) {
    if (!c) return 0.0;
    const int u = c->ufac();
    const double px = (double)N * (H / u) * (W / u);
    double f = 0.0;
    const int tail = 1 + c->nb * 15;
    for (int i = 0; i < (int)c->layers.size(); ++i) {
        double scale = 1.0;
        if (i == tail + 1) scale = 4.0;
        if (i >= tail + 2) scale = 16.0;
        f += conv_flops(c->layers[i], px * scale);
    }
    return f;
}

int nesr_preferred_batch(const nesr_ctx* c, int H, int W, int max_batch) {
    if (!c || H <= 0 || W <= 0 || max_batch <= 1) return 1;
    const int u = c->ufac();
    const int h = (H + u - 1) / u, w = (W + u - 1) / u;
    // workgroups per frame of the trunk convs (the kernel choice mirrors launch_conv3x3_bf16)
    const bool xl = c->dtype == NESR_DTYPE_BF16 && (long)h * w > 256L * 256L;
    static const int geo = [] { const char* e = getenv("NESR_XL_GEOMETRY"); return e ? atoi(e) : 4; }();
    const int xl_th = geo == 8 ? 32 : 16;
    const bool sp = c->dtype == NESR_DTYPE_F32_SPLIT;   // 8x32-px tiles, two workgroups per CU
    const long per = sp ? (long)((h + 7) / 8) * ((w + 31) / 32)
                        : xl ? (long)((h + xl_th - 1) / xl_th) * ((w + 31) / 32) : (long)((h + 7) / 8) * ((w + 15) / 16);
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    const long slots = (long)cus * ((xl && geo == 8) ? 1 : 2);   // co-resident workgroups
    auto eff = [&](int b) {
        const long wgs = per * b;
        return (double)wgs / (double)(((wgs + slots - 1) / slots) * slots);
    };
    double best_eff = 0.0;
    for (int b = 1; b <= max_batch; ++b) best_eff = eff(b) > best_eff ? eff(b) : best_eff;
    int best = 1;
    for (int b = 1; b <= max_batch; ++b)
        if (eff(b) >= best_eff - 0.02) best = b;   // the largest batch within 2 % of the best fill: fewest launches
    return best;
}

int nesr_set_concurrent(nesr_ctx* c, int concurrent) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    c->shared_device = concurrent ? 1 : 0;
    return NESR_OK;
}

int nesr_set_fused(nesr_ctx* c, int on) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    c->rdb_mode = on ? c->rdb_mode_init : 0;          // on: what the context was created with (NESR_RDB_FUSE / NESR_STRIP, default auto)
    c->strip_mode = on ? c->strip_mode_init : 0;
    return NESR_OK;
}

int nesr_fused_state(const nesr_ctx* c) {
    if (!c) return 0;
    const int on = c->dtype == NESR_DTYPE_BF16 ? c->strip_mode != 0 : (c->dtype == NESR_DTYPE_F32_SPLIT && c->rdb_mode != 0);
    return (on ? 1 : 0) | (c->fused_aborts << 1);
}

int nesr_debug_fault(nesr_ctx* c, int drop_workgroups) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    c->debug_drop = drop_workgroups > 0 ? drop_workgroups : 0;
    return NESR_OK;
}

int nesr_set_kernel_timing(nesr_ctx* c, int enable) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    c->timing = enable != 0;
    return NESR_OK;
}

int nesr_kernel_time_ms(nesr_ctx* c, double* total_ms, int64_t* launches, double* flops) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    double ms = 0.0;
    for (auto& pr : c->ev_pending) {
        HIP_TRY(hipEventSynchronize(pr.second));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, pr.first, pr.second));
        ms += t;
        c->ev_free.push_back(pr);
    }
    c->ev_pending.clear();
    if (total_ms) *total_ms = ms;
    if (launches) *launches = c->timed_launches;
    if (flops) *flops = c->timed_flops;
    c->timed_launches = 0;
    c->timed_flops = 0.0;
    return NESR_OK;
}

int nesr_check_status(nesr_ctx* c) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    if (c->last_sync) {
        unsigned flag = 0;
        HIP_TRY(hipMemcpy(&flag, c->last_sync, 4, hipMemcpyDeviceToHost));
        if (flag) return fail(NESR_ERR_HIP, "persistent trunk kernel aborted: a neighbour wait timed out (workgroups not co-resident?)");
    }
    return nesr_check_range(c, nullptr);
}

int nesr_check_range(nesr_ctx* c, void* stream) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    if (c->dtype != NESR_DTYPE_F32_SPLIT && !c->strip_used) return NESR_OK;   // the other forms compute in formats with f32's range
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(c->h_status, c->d_status, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->strip_used = false;
    if (c->h_status[2]) {
        const unsigned code = c->h_status[2];      // 1 | layer waited for << 8 | workgroup << 16
        HIP_TRY(hipMemsetAsync(c->d_status + 2, 0, 4, s));
        HIP_TRY(hipStreamSynchronize(s));
        c->h_status[2] = 0;
        c->strip_mode = 0;       // this context runs per-layer launches from now on (valid values; not the strip kernel's bits)
        ++c->fused_aborts;
        return fail(NESR_ERR_HIP, "the LDS-resident dense-block kernel gave up waiting for a neighbouring strip's edge column (workgroup " +
                                  std::to_string(code >> 16) + ", layer " + std::to_string((code >> 8) & 255u) +
                                  ": its workgroups were not all resident -- another process's persistent kernel shares the device?); the "
                                  "output of that forward is invalid; this context uses per-layer launches from now on (re-run the frame)");
    }
    if (c->dtype != NESR_DTYPE_F32_SPLIT) return NESR_OK;
    if (c->h_status[1]) {
        const unsigned code = c->h_status[1];      // 1 | chunk whose producer was waited for << 8 | tile << 16
        HIP_TRY(hipMemsetAsync(c->d_status, 0, 8, s));
        HIP_TRY(hipStreamSynchronize(s));
        c->h_status[0] = c->h_status[1] = 0;
        c->rdb_mode = 0;         // per-layer launches from now on: the same values bit for bit, no inter-workgroup waits
        ++c->fused_aborts;
        return fail(NESR_ERR_HIP, "the fused dense-block kernel gave up waiting for a neighbouring tile's progress word (tile " +
                                  std::to_string(code >> 16) + ", input chunk " + std::to_string((code >> 8) & 255u) +
                                  ": its workgroups were not all resident -- another persistent kernel shares the device?); the "
                                  "output of that forward is invalid; this context uses per-layer launches (the same bits) from now on: re-run the frame");
    }
    if (c->h_status[3] && !*c->h_status) {
        HIP_TRY(hipMemsetAsync(c->d_status + 3, 0, 4, s));
        HIP_TRY(hipStreamSynchronize(s));
        c->h_status[3] = 0;
        return fail(NESR_ERR_RANGE, "an EARLIER forward on this context (its result was never checked with nesr_check_range) met an input or "
                                    "activation of the f16-pair fp32 path that was non-finite or exceeded 65504 in magnitude: that forward's "
                                    "output was NaN / invalid; the latest forward's output is valid");
    }
    if (*c->h_status) {
        HIP_TRY(hipMemsetAsync(c->d_status, 0, 4, s));   // reported once; the next forward starts clean
        HIP_TRY(hipMemsetAsync(c->d_status + 3, 0, 4, s));
        HIP_TRY(hipStreamSynchronize(s));
        *c->h_status = 0;
        c->h_status[3] = 0;
        return fail(NESR_ERR_RANGE, "an input or activation of the f16-pair fp32 path was non-finite or exceeded 65504 in magnitude: "
                                    "the float output of that forward is NaN, an 8-bit output is invalid (use compute_dtype "
                                    "f32-winograd or f32-direct for such data)");
    }
    return NESR_OK;
}

void nesr_destroy(nesr_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& pr : c->ev_pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto& pr : c->ev_free) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (c->ws) (void)hipFree(c->ws);
    if (c->d_weights) (void)hipFree(c->d_weights);
    if (c->d_trunk) (void)hipFree(c->d_trunk);
    if (c->d_strip) (void)hipFree(c->d_strip);
    if (c->shard_buf) (void)hipFree(c->shard_buf);
    if (c->comm) (void)nesr_comm_destroy(c);
    free_strip_plans(c);
    lease_forget(c);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->h_status) (void)hipHostFree(c->h_status);
    delete c;
}

// ---- banded evaluation (exact multi-GPU mode: one row band of the frame per rank, SURVEY.md section 8(e) mode 2).
// The caller runs the stages in order and refreshes the apron rows of the feature map each stage reads
// (nesr_band_rows) with its neighbours' band rows in between; banded.py holds that protocol.
int nesr_band_begin(nesr_ctx* c, const void* x_dev, int C, int H, int W, void* stream) {
    if (!c || !x_dev) return fail(NESR_ERR_ARG, "null argument");
    c->band_valid = false;
    int rc = fw_setup(c, 1, C, H, W, c->band);
    if (rc) return rc;
    if ((rc = fw_first(c, c->band, static_cast<const float*>(x_dev), nullptr, 0, C, H, W, static_cast<hipStream_t>(stream)))) return rc;
    c->band_valid = true;
    return NESR_OK;
}

int nesr_band_rdb(nesr_ctx* c, int index, void* stream) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    if (!c->band_valid) return fail(NESR_ERR_STATE, "nesr_band_begin has not run (or a whole-frame forward reused the workspace)");
    if (index < 0 || index >= 3 * c->nb) return fail(NESR_ERR_ARG, "RDB index out of range");
    HIP_TRY(hipSetDevice(c->device));
    return fw_rdb(c, c->band, index / 3, index % 3, static_cast<hipStream_t>(stream));
}

int nesr_band_rdb_phase(nesr_ctx* c, int index, int phase, int top, int bottom, int edge_rows, void* stream) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    if (!c->band_valid) return fail(NESR_ERR_STATE, "nesr_band_begin has not run (or a whole-frame forward reused the workspace)");
    if (index < 0 || index >= 3 * c->nb) return fail(NESR_ERR_ARG, "RDB index out of range");
    if ((phase != 0 && phase != 1) || top < 0 || bottom < 0 || edge_rows < 0 || top + bottom > c->band.h)
        return fail(NESR_ERR_ARG, "bad phase / apron / edge rows");
    HIP_TRY(hipSetDevice(c->device));
    return fw_rdb(c, c->band, index / 3, index % 3, static_cast<hipStream_t>(stream), phase, top, bottom, edge_rows);
}

int nesr_band_pack_edges(nesr_ctx* c, int buffer, int top, int bottom, int nrows, void* top_dst, void* bottom_dst, void* stream) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    if (!c->band_valid) return fail(NESR_ERR_STATE, "nesr_band_begin has not run (or a whole-frame forward reused the workspace)");
    const int h = c->band.h;
    if (top < 0 || bottom < 0 || nrows < 0 || top + bottom + nrows > h) return fail(NESR_ERR_ARG, "bad apron / row count");
    int rc = NESR_OK;
    if (top_dst && (rc = nesr_band_rows(c, buffer, top, nrows, top_dst, 0, stream))) return rc;
    if (bottom_dst && (rc = nesr_band_rows(c, buffer, h - bottom - nrows, nrows, bottom_dst, 0, stream))) return rc;
    return NESR_OK;
}

int nesr_band_unpack_aprons(nesr_ctx* c, int buffer, int top, int bottom, int nrows, const void* top_src, const void* bottom_src, void* stream) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    if (!c->band_valid) return fail(NESR_ERR_STATE, "nesr_band_begin has not run (or a whole-frame forward reused the workspace)");
    const int h = c->band.h;
    if ((top_src && nrows > top) || (bottom_src && nrows > bottom) || nrows < 0) return fail(NESR_ERR_ARG, "more rows than the apron holds");
    int rc = NESR_OK;
    if (top_src && (rc = nesr_band_rows(c, buffer, top - nrows, nrows, const_cast<void*>(top_src), 1, stream))) return rc;
    if (bottom_src && (rc = nesr_band_rows(c, buffer, h - bottom, nrows, const_cast<void*>(bottom_src), 1, stream))) return rc;
    return NESR_OK;
}

int nesr_band_tail(nesr_ctx* c, void* y_dev, void* stream) {
    if (!c || !y_dev) return fail(NESR_ERR_ARG, "null argument");
    if (!c->band_valid) return fail(NESR_ERR_STATE, "nesr_band_begin has not run (or a whole-frame forward reused the workspace)");
    HIP_TRY(hipSetDevice(c->device));
    return fw_tail(c, c->band, static_cast<float*>(y_dev), nullptr, 0, 0, static_cast<hipStream_t>(stream));
}

size_t nesr_band_row_bytes(const nesr_ctx* c) {
    if (!c || !c->band_valid) return 0;
    return (size_t)c->band.w * c->nf * c->esize();
}

int nesr_band_rows(nesr_ctx* c, int buffer, int row0, int nrows, void* staging_dev, int write, void* stream) {
    if (!c || !staging_dev) return fail(NESR_ERR_ARG, "null argument");
    if (!c->band_valid) return fail(NESR_ERR_STATE, "nesr_band_begin has not run (or a whole-frame forward reused the workspace)");
    const FwState& F = c->band;
    if (buffer < 0 || buffer > 3 || row0 < 0 || nrows < 0 || row0 + nrows > F.h) return fail(NESR_ERR_ARG, "bad buffer / row range");
    if (nrows == 0) return NESR_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* base = buffer < 3 ? F.buf[buffer] : c->ws + F.L.f;
    const Map& m = buffer < 3 ? F.m_t : F.m_f;
    char* stg = static_cast<char*>(staging_dev);
    const int kind = c->kind();
    if (kind == 0) {
        // NHWC f32: channels [0, nf) of every pixel of the rows; the dense-block buffers have ct channels per pixel
        const size_t spitch = (size_t)m.pix * 4, width = (size_t)c->nf * 4, rows = (size_t)nrows * F.w;
        char* src = base + (size_t)row0 * F.w * spitch;
        if (write) HIP_TRY(hipMemcpy2DAsync(src, spitch, stg, width, width, rows, hipMemcpyDeviceToDevice, s));
        else HIP_TRY(hipMemcpy2DAsync(stg, width, src, spitch, width, rows, hipMemcpyDeviceToDevice, s));
        return NESR_OK;
    }
    // channel-blocked: the rows of one 16-channel chunk are one contiguous span; staging = [chunk][rows][w][pixel bytes]
    const size_t pixbytes = (size_t)m.pix * 2, span = (size_t)nrows * F.w * pixbytes;
    for (int ch = 0; ch < c->nf / 16; ++ch) {
        char* src = base + (size_t)ch * (size_t)m.chunk * 2 + (size_t)row0 * F.w * pixbytes;
        char* dst = stg + (size_t)ch * span;
        if (write) HIP_TRY(hipMemcpyAsync(src, dst, span, hipMemcpyDeviceToDevice, s));
        else HIP_TRY(hipMemcpyAsync(dst, src, span, hipMemcpyDeviceToDevice, s));
    }
    return NESR_OK;
}

int nesr_cut_tiles_u8(int device_id, const uint8_t* frame_hwc_dev, int H, int W, int flip_rgb, int through_fp16, const int* windows, int n, int Hs, int Ws,
                      float* tiles_nchw_dev, void* stream) {
    if (!frame_hwc_dev || !windows || !tiles_nchw_dev) return fail(NESR_ERR_ARG, "null argument");
    if (n < 1 || n > TILE_IO_MAX || Hs < 1 || Ws < 1) return fail(NESR_ERR_ARG, "nesr_cut_tiles_u8: 1.." + std::to_string(TILE_IO_MAX) + " tiles per call");
    TileIo t;
    std::memset(&t, 0, sizeof(t));
    for (int i = 0; i < n; ++i) {
        const int y0 = windows[4 * i], x0 = windows[4 * i + 1], h = windows[4 * i + 2], w = windows[4 * i + 3];
        if (y0 < 0 || x0 < 0 || h < 1 || w < 1 || y0 + h > H || x0 + w > W || h > Hs || w > Ws)
            return fail(NESR_ERR_ARG, "nesr_cut_tiles_u8: window " + std::to_string(i) + " outside the frame or larger than a slot");
        t.desc[8 * i] = y0; t.desc[8 * i + 1] = x0; t.desc[8 * i + 2] = h; t.desc[8 * i + 3] = w;
    }
    t.frame = const_cast<uint8_t*>(frame_hwc_dev); t.frame_w = W; t.tiles = tiles_nchw_dev; t.Hs = Hs; t.Ws = Ws; t.flip = flip_rgb ? 1 : 0;
    t.round = through_fp16 ? 1 : 0;
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(launch_cut_tiles(t, n, Hs, Ws, static_cast<hipStream_t>(stream)));
    return NESR_OK;
}

int nesr_paste_tiles_u8(int device_id, const float* tiles_nchw_dev, int n, int Hs, int Ws, const int64_t* desc, uint8_t* dst_dev, size_t dst_bytes,
                        int flip_rgb, int round_mode, int through_fp16, void* stream) {
    if (!tiles_nchw_dev || !desc || !dst_dev) return fail(NESR_ERR_ARG, "null argument");
    if (n < 1 || n > TILE_IO_MAX || Hs < 1 || Ws < 1) return fail(NESR_ERR_ARG, "nesr_paste_tiles_u8: 1.." + std::to_string(TILE_IO_MAX) + " tiles per call");
    TileIo t;
    std::memset(&t, 0, sizeof(t));
    int maxh = 0, maxw = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t* d = desc + 6 * i;      // crop y, crop x, h, w, destination offset (bytes), row pitch (bytes)
        if (d[0] < 0 || d[1] < 0 || d[2] < 1 || d[3] < 1 || d[0] + d[2] > Hs || d[1] + d[3] > Ws || d[4] < 0 || d[5] < d[3] * 3 ||
            (uint64_t)d[4] + (uint64_t)(d[2] - 1) * (uint64_t)d[5] + (uint64_t)d[3] * 3 > dst_bytes)
            return fail(NESR_ERR_ARG, "nesr_paste_tiles_u8: tile " + std::to_string(i) + ": crop outside its slot or destination outside the buffer");
        int* o = t.desc + 8 * i;
        o[0] = (int)d[0]; o[1] = (int)d[1]; o[2] = (int)d[2]; o[3] = (int)d[3]; o[4] = (int)d[5];
        o[5] = (int)(uint32_t)((uint64_t)d[4] & 0xffffffffull); o[6] = (int)(uint32_t)((uint64_t)d[4] >> 32);
        maxh = std::max(maxh, (int)d[2]); maxw = std::max(maxw, (int)d[3]);
    }
    t.frame = dst_dev; t.tiles = const_cast<float*>(tiles_nchw_dev); t.Hs = Hs; t.Ws = Ws; t.flip = flip_rgb ? 1 : 0;
    t.round = (round_mode == NESR_ROUND_NEAREST ? 1 : 0) | (through_fp16 ? 2 : 0);
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(launch_paste_tiles(t, n, maxh, maxw, static_cast<hipStream_t>(stream)));
    return NESR_OK;
}

int nesr_nl_means_u8(int device_id, const uint8_t* planes_dev, int C, int H, int W, int template_size, int search_size, const int* weights_dev, int nbins,
                     uint8_t* out_dev, void* stream) {
    if (!planes_dev || !weights_dev || !out_dev) return fail(NESR_ERR_ARG, "null argument");
    if (template_size != 7 || search_size != 21) return fail(NESR_ERR_ARG, "nesr_nl_means_u8: template 7 / search 21 (what nesr/nesr.py:674 passes)");
    if (C < 1 || C > 3 || H < 1 || W < 1 || nbins < 1) return fail(NESR_ERR_ARG, "nesr_nl_means_u8: 1..3 planes, a non-empty image and table");
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(launch_nl_means(planes_dev, C, H, W, weights_dev, nbins, 6 /* 49 template pixels -> next power of two 64 */, out_dev, static_cast<hipStream_t>(stream)));
    return NESR_OK;
}

int nesr_clahe_u8(int device_id, const uint8_t* gray_dev, int H, int W, double clip_limit, int grid_x, int grid_y, float* lut_dev, uint8_t* out_dev, void* stream) {
    if (!gray_dev || !lut_dev || !out_dev) return fail(NESR_ERR_ARG, "null argument");
    if (H < 1 || W < 1 || grid_x < 1 || grid_y < 1 || grid_x * grid_y > 4096 || !(clip_limit > 0.0)) return fail(NESR_ERR_ARG, "nesr_clahe_u8: non-empty image, grid and clip limit");
    // clahe.cpp: the image is used as it is only when BOTH sides divide by the grid; otherwise both are padded
    const int ph = (H % grid_y || W % grid_x) ? grid_y - H % grid_y : 0, pw = (H % grid_y || W % grid_x) ? grid_x - W % grid_x : 0;
    const int th = (H + ph) / grid_y, tw = (W + pw) / grid_x;
    const long long area = (long long)th * tw;
    if (area > (1ll << 30)) return fail(NESR_ERR_ARG, "nesr_clahe_u8: tile too large");
    int clip = (int)(clip_limit * (double)area / 256.0);
    clip = clip < 1 ? 1 : clip;
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(launch_clahe(gray_dev, H, W, grid_x, grid_y, th, tw, clip, (float)(255.0 / (double)area), 1.0f / (float)th, 1.0f / (float)tw, lut_dev, out_dev,
                         static_cast<hipStream_t>(stream)));
    return NESR_OK;
}

int nesr_conv3x3(int device_id, int dtype, const void* x_dev, int N, int Cin, int H, int W, const float* w_host,
                 const float* b_host, int Cout, int lrelu, int upsample, void* y_dev, void* stream) {
    if (!x_dev || !w_host || !b_host || !y_dev) return fail(NESR_ERR_ARG, "null argument");
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout > 64) return fail(NESR_ERR_ARG, "bad shape (Cout <= 64)");
    if (dtype != NESR_DTYPE_F32 && dtype != NESR_DTYPE_BF16 && dtype != NESR_DTYPE_F32_WINOGRAD && dtype != NESR_DTYPE_F32_SPLIT)
        return fail(NESR_ERR_ARG, "bad dtype");
    const bool wino = dtype == NESR_DTYPE_F32_WINOGRAD;
    const bool sp = dtype == NESR_DTYPE_F32_SPLIT;
    const int kind = sp ? 2 : (dtype == NESR_DTYPE_BF16 ? 1 : 0);
    HIP_TRY(hipSetDevice(device_id));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool bf = dtype == NESR_DTYPE_BF16;
    const size_t es = bf ? 2 : 4;
    const int cin_p = round_up(Cin, (bf || sp) ? 16 : 8), cout_p = round_up(Cout, 32);
    const int up = upsample ? 1 : 0;
    const int ho = H << up, wo = W << up;
    const size_t we = sp ? packed_weight_elems_f16x2(cin_p, cout_p) / 2 : bf ? packed_weight_elems_bf16(cin_p, cout_p)
                         : (wino ? packed_weight_elems_wino_f32(cin_p, cout_p) : packed_weight_elems_f32(cin_p, cout_p));
    std::vector<char> hw(we * es);
    if (sp)
        pack_weights_f16x2(w_host, Cout, Cin, cin_p, cout_p, reinterpret_cast<uint16_t*>(hw.data()));
    else if (bf)
        pack_weights_bf16(w_host, Cout, Cin, cin_p, cout_p, reinterpret_cast<uint16_t*>(hw.data()));
    else if (wino)
        pack_weights_wino_f32(w_host, Cout, Cin, cin_p, cout_p, reinterpret_cast<float*>(hw.data()));
    else
        pack_weights_f32(w_host, Cout, Cin, cin_p, cout_p, reinterpret_cast<float*>(hw.data()));
    std::vector<float> hb(cout_p, 0.f);
    std::memcpy(hb.data(), b_host, (size_t)Cout * 4);
    // device scratch of this one call; freed on every return path
    struct Scratch {
        std::vector<void*> ptrs;
        ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
        hipError_t take(void** p, size_t bytes) {
            const hipError_t e = hipMalloc(p, bytes);
            if (e == hipSuccess) ptrs.push_back(*p);
            return e;
        }
    } scratch;
    char *d_w = nullptr, *d_in = nullptr, *d_out = nullptr, *d_zero = nullptr;
    float* d_b = nullptr;
    const size_t in_bytes = (size_t)N * H * W * cin_p * es, out_bytes = (size_t)N * ho * wo * cout_p * es;
    HIP_TRY(scratch.take((void**)&d_w, hw.size()));
    HIP_TRY(scratch.take((void**)&d_b, hb.size() * 4));
    HIP_TRY(scratch.take((void**)&d_in, in_bytes));
    HIP_TRY(scratch.take((void**)&d_out, out_bytes));
    HIP_TRY(scratch.take((void**)&d_zero, 256));
    HIP_TRY(hipMemset(d_zero, 0, 256));
    unsigned* d_status = reinterpret_cast<unsigned*>(d_zero + 128);   // the upper half of the zero page is never a DMA source (>= 16 B needed)
    HIP_TRY(hipMemcpy(d_w, hw.data(), hw.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    PackArgs p;
    std::memset(&p, 0, sizeof(p));
    const Map mi = make_map(kind, cin_p, (size_t)N * H * W), mo = make_map(kind, cout_p, (size_t)N * ho * wo);
    p.src = x_dev; p.n = N; p.c = Cin; p.hin = H; p.win = W; p.unshuffle = 1; p.dst = d_in; p.dst_map = mi; p.cp = cin_p; p.bf16 = kind;
    p.status = sp ? d_status : nullptr;
    HIP_TRY(launch_pack_input(p, s));
    ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = d_in; a.in_map = mi; a.in_h = H; a.in_w = W; a.up = up; a.cin = cin_p;
    a.w = d_w; a.bias = d_b; a.coutp = cout_p;
    a.n = N; a.h = ho; a.w_ = wo;
    a.out = d_out; a.out_map = mo; a.out_coff = 0;
    a.lrelu = lrelu ? 1 : 0; a.s1 = a.s2 = 1.f;
    a.zeros = d_zero;
    a.status = sp ? d_status : nullptr;
    HIP_TRY(sp ? launch_conv3x3_f16x2(a, s) : bf ? launch_conv3x3_bf16(a, s) : (wino ? launch_conv3x3_wino_f32(a, s) : launch_conv3x3_f32(a, s)));
    HIP_TRY(launch_nhwc_to_nchw(d_out, kind, mo, N, Cout, ho, wo, static_cast<float*>(y_dev), s));
    HIP_TRY(hipStreamSynchronize(s));
    if (sp) {
        unsigned flag = 0;
        HIP_TRY(hipMemcpy(&flag, d_status, 4, hipMemcpyDeviceToHost));
        if (flag) return fail(NESR_ERR_RANGE, "input or output of the layer was non-finite or exceeded 65504 in magnitude (f16-pair form)");
        for (size_t i = 0; i < (size_t)Cout * Cin * 9; ++i)
            if (!(std::fabs(w_host[i]) <= 65504.f)) return fail(NESR_ERR_RANGE, "weight does not fit the f16-pair form (|w| > 65504 or non-finite)");
    }
    return NESR_OK;
}

}  // extern "C"

// ======================================================================================================================
// Sharded frames below Python (SURVEY.md section 8(b), 8(e) mode 1): one process per GPU, the tiles of upstream's tile grid
// dealt to the ranks, the input frame row-scattered; a rank fetches the rows its tiles read beyond its own band from the owning
// ranks (RCCL point to point over xGMI: grouped ncclSend / ncclRecv), evaluates its tiles (cut -> ragged forward -> paste) and
// sends their quantised centres to rank 0.  No collective: tiles are independent network evaluations
// (RealESRGANer.tile_process, standalone/direct_esrgan.py:118-127,148).  neural_enhanced_super_resolution_amd/sharded.py is the same
// protocol over torch.distributed; nesr_shard_plan is tested against it plan for plan.
namespace {

// RCCL is loaded on first use (librccl.so is 570 MB; a single-GPU user never pays for it, and the library loads without it)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /* ncclUniqueId by value: 128 bytes */ struct Id128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
struct Id128 { char b[128]; };
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.lib) return NESR_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(NESR_ERR_STATE, std::string("RCCL is not loadable (librccl.so): ") + dlerror());
#define RSYM(field, name)                                                                    \
    *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, name);                               \
    if (!g_rccl.field) return fail(NESR_ERR_STATE, std::string("librccl.so lacks ") + name);
    RSYM(GetUniqueId, "ncclGetUniqueId")
    RSYM(CommInitRank, "ncclCommInitRank")
    RSYM(CommDestroy, "ncclCommDestroy")
    RSYM(Send, "ncclSend")
    RSYM(Recv, "ncclRecv")
    RSYM(GroupStart, "ncclGroupStart")
    RSYM(GroupEnd, "ncclGroupEnd")
    RSYM(GetErrorString, "ncclGetErrorString")
#undef RSYM
    g_rccl.lib = h;
    return NESR_OK;
}
#define RCCL_TRY(expr)                                                                                              \
    do {                                                                                                            \
        const int r__ = (expr);                                                                                     \
        if (r__ != 0) return fail(NESR_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r__));          \
    } while (0)
constexpr int NCCL_UINT8 = 1;      // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)

struct ShardTile { int inp[4], out[4], crop[4], owner; long area() const { return (long)(inp[1] - inp[0]) * (inp[3] - inp[2]); } };

// RealESRGANer.tile_grid + sharded.plan_tiles: upstream's windows in its order; contiguous runs of tiles per rank, balanced by padded
// input area (a tile goes to the next rank once its midpoint passes the rank's share)
std::vector<ShardTile> shard_tiles(int H, int W, int s, int tile, int pad, int world) {
    std::vector<ShardTile> v;
    if (tile <= 0) {
        ShardTile t{{0, H, 0, W}, {0, H * s, 0, W * s}, {0, H * s, 0, W * s}, 0};
        v.push_back(t);
    } else {
        const int tx = (W + tile - 1) / tile, ty = (H + tile - 1) / tile;
        for (int y = 0; y < ty; ++y)
            for (int x = 0; x < tx; ++x) {
                const int ix0 = x * tile, iy0 = y * tile, ix1 = std::min(ix0 + tile, W), iy1 = std::min(iy0 + tile, H);
                const int px0 = std::max(ix0 - pad, 0), px1 = std::min(ix1 + pad, W), py0 = std::max(iy0 - pad, 0), py1 = std::min(iy1 + pad, H);
                const int cx0 = (ix0 - px0) * s, cy0 = (iy0 - py0) * s;
                ShardTile t{{py0, py1, px0, px1}, {iy0 * s, iy1 * s, ix0 * s, ix1 * s}, {cy0, cy0 + (iy1 - iy0) * s, cx0, cx0 + (ix1 - ix0) * s}, 0};
                v.push_back(t);
            }
    }
    double total = 0;
    for (auto& t : v) total += (double)t.area();
    double acc = 0;
    int r = 0;
    for (auto& t : v) {
        while (r < world - 1 && acc + (double)t.area() / 2 > (double)(r + 1) * total / world) ++r;
        t.owner = r;
        acc += (double)t.area();
    }
    return v;
}
void rows_needed(const std::vector<ShardTile>& v, int rank, int& n0, int& n1) {
    n0 = n1 = 0;
    bool any = false;
    for (const auto& t : v)
        if (t.owner == rank) {
            n0 = any ? std::min(n0, t.inp[0]) : t.inp[0];
            n1 = any ? std::max(n1, t.inp[1]) : t.inp[1];
            any = true;
        }
}
struct RowMove { int src, dst, lo, hi; };
std::vector<RowMove> shard_exchange(const std::vector<ShardTile>& v, int world, int H) {
    std::vector<RowMove> plan;
    for (int d = 0; d < world; ++d) {
        int n0, n1;
        rows_needed(v, d, n0, n1);
        for (int sr = 0; sr < world; ++sr) {
            if (sr == d) continue;
            const int b0 = (int)((long)sr * H / world), b1 = (int)((long)(sr + 1) * H / world);
            const int lo = std::max(n0, b0), hi = std::min(n1, b1);
            if (lo < hi) plan.push_back({sr, d, lo, hi});
        }
    }
    return plan;
}

int ensure_shard_buf(nesr_ctx* c, size_t bytes) {
    if (bytes <= c->shard_bytes) return NESR_OK;
    if (c->shard_buf) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->shard_buf)); c->shard_buf = nullptr; c->shard_bytes = 0; }
    void* p = nullptr;
    const hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(NESR_ERR_NOMEM, "hipMalloc(shard scratch " + std::to_string(bytes) + " B): " + hipGetErrorString(e));
    c->shard_buf = static_cast<char*>(p);
    c->shard_bytes = bytes;
    return NESR_OK;
}

}  // namespace

extern "C" {

int nesr_shard_plan(int H, int W, int scale, int tile, int tile_pad, int nranks, int* tiles13, int cap_tiles, int* ntiles, int* moves4, int cap_moves,
                    int* nmoves) {
    if (H < 1 || W < 1 || scale < 1 || tile < 0 || tile_pad < 0 || nranks < 1 || !ntiles || !nmoves) return fail(NESR_ERR_ARG, "nesr_shard_plan: bad argument");
    const auto v = shard_tiles(H, W, scale, tile, tile_pad, nranks);
    const auto m = shard_exchange(v, nranks, H);
    *ntiles = (int)v.size();
    *nmoves = (int)m.size();
    if (tiles13 && (int)v.size() <= cap_tiles)
        for (size_t i = 0; i < v.size(); ++i) {
            for (int k = 0; k < 4; ++k) { tiles13[13 * i + k] = v[i].inp[k]; tiles13[13 * i + 4 + k] = v[i].out[k]; tiles13[13 * i + 8 + k] = v[i].crop[k]; }
            tiles13[13 * i + 12] = v[i].owner;
        }
    if (moves4 && (int)m.size() <= cap_moves)
        for (size_t i = 0; i < m.size(); ++i) { moves4[4 * i] = m[i].src; moves4[4 * i + 1] = m[i].dst; moves4[4 * i + 2] = m[i].lo; moves4[4 * i + 3] = m[i].hi; }
    return NESR_OK;
}

int nesr_comm_unique_id(void* id128) {
    if (!id128) return fail(NESR_ERR_ARG, "null id");
    int rc = rccl_load();
    if (rc) return rc;
    RCCL_TRY(g_rccl.GetUniqueId(id128));
    return NESR_OK;
}

int nesr_comm_init(nesr_ctx* c, int rank, int nranks, const void* id128) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(NESR_ERR_ARG, "nesr_comm_init: bad argument");
    int rc = rccl_load();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (c->comm) { RCCL_TRY(g_rccl.CommDestroy(c->comm)); c->comm = nullptr; }
    Id128 id;
    std::memcpy(id.b, id128, 128);
    RCCL_TRY(g_rccl.CommInitRank(&c->comm, nranks, id, rank));
    c->comm_rank = rank;
    c->comm_nranks = nranks;
    return NESR_OK;
}

int nesr_comm_destroy(nesr_ctx* c) {
    if (!c) return fail(NESR_ERR_ARG, "null ctx");
    if (c->comm) { RCCL_TRY(g_rccl.CommDestroy(c->comm)); c->comm = nullptr; }
    c->comm_rank = 0;
    c->comm_nranks = 1;
    return NESR_OK;
}

int nesr_forward_sharded_u8(nesr_ctx* c, const uint8_t* band_dev, int H, int W, int tile, int tile_pad, int through_fp16, uint8_t* out_dev, void* stream) {
    if (!c || !band_dev) return fail(NESR_ERR_ARG, "null argument");
    const int world = c->comm ? c->comm_nranks : 1, rank = c->comm ? c->comm_rank : 0;
    const int u = c->ufac(), s = 4 / u;
    if (c->cin0 != 3 * u * u || c->nout != 3) return fail(NESR_ERR_ARG, "nesr_forward_sharded_u8 needs a 3-channel-in / 3-channel-out network");
    if (H < 1 || W < 1 || H % u || W % u || tile < 0 || tile_pad < 0) return fail(NESR_ERR_ARG, "nesr_forward_sharded_u8: frame sides must be multiples of the unshuffle factor");
    if (rank == 0 && !out_dev) return fail(NESR_ERR_ARG, "rank 0 needs the output canvas");
    if (c->dtype != NESR_DTYPE_BF16) return fail(NESR_ERR_ARG, "nesr_forward_sharded_u8: compute dtype bf16 (ragged tile batches)");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const auto tiles = shard_tiles(H, W, s, tile, tile_pad, world);
    const auto moves = shard_exchange(tiles, world, H);
    int n0, n1;
    rows_needed(tiles, rank, n0, n1);
    const int b0 = (int)((long)rank * H / world), b1 = (int)((long)(rank + 1) * H / world);
    std::vector<const ShardTile*> mine;
    int hs = 0, ws = 0;
    size_t packed = 0;
    for (const auto& t : tiles)
        if (t.owner == rank) {
            mine.push_back(&t);
            hs = std::max(hs, t.inp[1] - t.inp[0]);
            ws = std::max(ws, t.inp[3] - t.inp[2]);
            packed += (size_t)(t.out[1] - t.out[0]) * (t.out[3] - t.out[2]) * 3;
        }
    size_t remote = 0;      // rank 0: bytes of the other ranks' tiles
    if (rank == 0)
        for (const auto& t : tiles)
            if (t.owner != 0) remote += (size_t)(t.out[1] - t.out[0]) * (t.out[3] - t.out[2]) * 3;
    // scratch: [local rows | x tiles | y tiles | packed results (ranks > 0) or remote staging (rank 0)]
    const size_t row_bytes = (size_t)W * 3;
    const size_t local_bytes = align_up((size_t)std::max(n1 - n0, 0) * row_bytes, 256);
    const int batch = (int)std::min<size_t>(mine.size(), std::min(TILE_IO_MAX, (int)RAG_MAX));
    const size_t x_bytes = align_up((size_t)batch * 3 * hs * ws * 4, 256), y_bytes = align_up((size_t)batch * 3 * hs * s * ws * s * 4, 256);
    const size_t tail_bytes = align_up(rank == 0 ? remote : packed, 256);
    int rc = ensure_shard_buf(c, local_bytes + x_bytes + y_bytes + tail_bytes + 256);
    if (rc) return rc;
    uint8_t* local = reinterpret_cast<uint8_t*>(c->shard_buf);
    float* xt = reinterpret_cast<float*>(c->shard_buf + local_bytes);
    float* yt = reinterpret_cast<float*>(c->shard_buf + local_bytes + x_bytes);
    uint8_t* tailb = reinterpret_cast<uint8_t*>(c->shard_buf + local_bytes + x_bytes + y_bytes);
    // ---- own rows, then the rows of other bands (overlap rows and whatever the balanced assignment shifts across a band edge)
    {
        const int lo = std::max(n0, b0), hi = std::min(n1, b1);
        if (lo < hi) HIP_TRY(hipMemcpyAsync(local + (size_t)(lo - n0) * row_bytes, band_dev + (size_t)(lo - b0) * row_bytes, (size_t)(hi - lo) * row_bytes, hipMemcpyDeviceToDevice, st));
    }
    if (world > 1) {
        RCCL_TRY(g_rccl.GroupStart());
        for (const auto& m : moves) {
            if (m.src == rank) RCCL_TRY(g_rccl.Send(band_dev + (size_t)(m.lo - b0) * row_bytes, (size_t)(m.hi - m.lo) * row_bytes, NCCL_UINT8, m.dst, c->comm, st));
            else if (m.dst == rank) RCCL_TRY(g_rccl.Recv(local + (size_t)(m.lo - n0) * row_bytes, (size_t)(m.hi - m.lo) * row_bytes, NCCL_UINT8, m.src, c->comm, st));
        }
        RCCL_TRY(g_rccl.GroupEnd());
    }
    // ---- this rank's tiles: cut -> ragged forward -> paste (rank 0: into the canvas; others: packed, tile after tile)
    size_t poff = 0;
    for (size_t i0 = 0; i0 < mine.size(); i0 += batch) {
        const int n = (int)std::min<size_t>(batch, mine.size() - i0);
        TileIo cut, pst;
        std::memset(&cut, 0, sizeof(cut));
        std::memset(&pst, 0, sizeof(pst));
        std::vector<int> hw(2 * (size_t)n);
        int bh = 0, bw = 0, ph = 0, pw = 0;
        for (int i = 0; i < n; ++i) {
            const ShardTile& t = *mine[i0 + i];
            bh = std::max(bh, t.inp[1] - t.inp[0]);
            bw = std::max(bw, t.inp[3] - t.inp[2]);
        }
        for (int i = 0; i < n; ++i) {
            const ShardTile& t = *mine[i0 + i];
            int* d = cut.desc + 8 * i;
            d[0] = t.inp[0] - n0; d[1] = t.inp[2]; d[2] = t.inp[1] - t.inp[0]; d[3] = t.inp[3] - t.inp[2];
            hw[2 * i] = d[2]; hw[2 * i + 1] = d[3];
            int* o = pst.desc + 8 * i;
            o[0] = t.crop[0]; o[1] = t.crop[2]; o[2] = t.crop[1] - t.crop[0]; o[3] = t.crop[3] - t.crop[2];
            size_t off;
            if (rank == 0) { off = ((size_t)t.out[0] * W * s + t.out[2]) * 3; o[4] = W * s * 3; }
            else { off = poff; o[4] = o[3] * 3; poff += (size_t)o[2] * o[3] * 3; }
            o[5] = (int)(uint32_t)(off & 0xffffffffull); o[6] = (int)(uint32_t)(off >> 32);
            ph = std::max(ph, o[2]); pw = std::max(pw, o[3]);
        }
        cut.frame = local; cut.frame_w = W; cut.tiles = xt; cut.Hs = bh; cut.Ws = bw; cut.flip = 1; cut.round = through_fp16 ? 1 : 0;
        HIP_TRY(launch_cut_tiles(cut, n, bh, bw, st));
        if ((rc = nesr_forward_ragged(c, xt, n, 3, bh, bw, hw.data(), yt, stream))) return rc;
        pst.frame = rank == 0 ? out_dev : tailb; pst.tiles = yt; pst.Hs = bh * s; pst.Ws = bw * s; pst.flip = 1; pst.round = 1 | (through_fp16 ? 2 : 0);
        HIP_TRY(launch_paste_tiles(pst, n, ph, pw, st));
    }
    // ---- gather on rank 0: one message per rank (its tiles packed in tile order), scattered into the canvas
    if (world > 1) {
        std::vector<size_t> rank_bytes(world, 0), rank_off(world, 0);
        for (const auto& t : tiles) rank_bytes[t.owner] += (size_t)(t.out[1] - t.out[0]) * (t.out[3] - t.out[2]) * 3;
        size_t o = 0;
        for (int r = 1; r < world; ++r) { rank_off[r] = o; o += rank_bytes[r]; }
        RCCL_TRY(g_rccl.GroupStart());
        if (rank == 0) {
            for (int r = 1; r < world; ++r)
                if (rank_bytes[r]) RCCL_TRY(g_rccl.Recv(tailb + rank_off[r], rank_bytes[r], NCCL_UINT8, r, c->comm, st));
        } else if (rank_bytes[rank]) {
            RCCL_TRY(g_rccl.Send(tailb, rank_bytes[rank], NCCL_UINT8, 0, c->comm, st));
        }
        RCCL_TRY(g_rccl.GroupEnd());
        if (rank == 0) {
            std::vector<size_t> cur(rank_off);
            for (const auto& t : tiles) {
                if (t.owner == 0) continue;
                const size_t th = t.out[1] - t.out[0], tw3 = (size_t)(t.out[3] - t.out[2]) * 3;
                HIP_TRY(hipMemcpy2DAsync(out_dev + ((size_t)t.out[0] * W * s + t.out[2]) * 3, (size_t)W * s * 3, tailb + cur[t.owner], tw3, tw3, th,
                                         hipMemcpyDeviceToDevice, st));
                cur[t.owner] += th * tw3;
            }
        }
    }
    return NESR_OK;
}

}  // extern "C"
