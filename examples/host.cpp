// A C++ host on the C ABI of libnesr_hip.so, with no Python and no torch in the process: what a non-Python
// integration of this path links against (include/nesr_hip.h).  Builds a 2-block x2plus network from seeded
// weights, runs one 64x96 frame through nesr_forward and once more through the fused u8 entry, and checks the
// two against each other; then the same weights as a bf16 context evaluate the frame TILED through
// nesr_forward_sharded_u8 (the entry a multi-GPU C host calls on every rank; here one rank, no communicator) and the
// plan four ranks would follow is printed (nesr_shard_plan).
//   hipcc -O2 --offload-arch=gfx950 -I include examples/host.cpp -o build/nesr_host -ldl && build/nesr_host path/to/libnesr_hip.so
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "nesr_hip.h"

#define LOAD(name) auto p_##name = reinterpret_cast<decltype(&name)>(dlsym(lib, #name)); if (!p_##name) { std::fprintf(stderr, "missing %s\n", #name); return 2; }
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 5; } } while (0)
#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, p_nesr_last_error()); return 3; } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float uniform() {   // splitmix64 -> [-1, 1)
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)((double)(z >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s libnesr_hip.so\n", argv[0]); return 1; }
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { std::fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(nesr_create) LOAD(nesr_load_weight) LOAD(nesr_finalize_weights) LOAD(nesr_forward) LOAD(nesr_forward_u8)
    LOAD(nesr_destroy) LOAD(nesr_last_error) LOAD(nesr_version) LOAD(nesr_num_tensors) LOAD(nesr_forward_sharded_u8) LOAD(nesr_shard_plan)
    LOAD(nesr_check_status)
    std::printf("%s\n", p_nesr_version());

    const int nf = 64, gc = 32, nb = 2, H = 64, W = 96;
    nesr_ctx *ctx = nullptr, *ctx16 = nullptr;
    CHECK(p_nesr_create(&ctx, 0, 12, 2, nf, nb, gc, 3, NESR_DTYPE_F32_SPLIT));
    CHECK(p_nesr_create(&ctx16, 0, 12, 2, nf, nb, gc, 3, NESR_DTYPE_BF16));
    auto conv = [&](const std::string& name, int cin, int cout) -> int {
        std::vector<float> w((size_t)cout * cin * 9), b(cout);
        const float sc = 0.5f * std::sqrt(2.0f / (cin * 9.0f));
        for (auto& v : w) v = uniform() * sc;
        for (auto& v : b) v = uniform() * 0.01f;
        const int64_t ws[4] = {cout, cin, 3, 3}, bs[1] = {cout};
        for (nesr_ctx* c : {ctx, ctx16}) {
            CHECK(p_nesr_load_weight(c, (name + ".weight").c_str(), w.data(), ws, 4));
            CHECK(p_nesr_load_weight(c, (name + ".bias").c_str(), b.data(), bs, 1));
        }
        return 0;
    };
    if (conv("conv_first", 12, nf)) return 3;
    for (int b = 0; b < nb; ++b)
        for (int r = 1; r <= 3; ++r) {
            const std::string pre = "body." + std::to_string(b) + ".rdb" + std::to_string(r) + ".conv";
            for (int k = 1; k <= 4; ++k)
                if (conv(pre + std::to_string(k), nf + (k - 1) * gc, gc)) return 3;
            if (conv(pre + "5", nf + 4 * gc, nf)) return 3;
        }
    for (const char* n : {"conv_body", "conv_up1", "conv_up2", "conv_hr"})
        if (conv(n, nf, nf)) return 3;
    if (conv("conv_last", nf, 3)) return 3;
    CHECK(p_nesr_finalize_weights(ctx));
    CHECK(p_nesr_finalize_weights(ctx16));
    std::printf("tensors loaded: %d\n", p_nesr_num_tensors(ctx));

    // one BGR u8 frame; the float entry gets RGB / 255 in NCHW, as RealESRGANer.enhance prepares it
    std::vector<uint8_t> img((size_t)H * W * 3);
    for (auto& v : img) v = (uint8_t)((uniform() * 0.5f + 0.5f) * 255.0f);
    std::vector<float> x((size_t)3 * H * W);
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < H * W; ++i) x[(size_t)c * H * W + i] = (float)img[(size_t)i * 3 + (2 - c)] / 255.0f;
    float *dx, *dy; uint8_t *dimg, *dout;
    const size_t out_px = (size_t)4 * H * W;
    uint8_t* dtiled;
    HIPCHK(hipMalloc(&dx, x.size() * 4)); HIPCHK(hipMalloc(&dy, out_px * 3 * 4)); HIPCHK(hipMalloc(&dimg, img.size())); HIPCHK(hipMalloc(&dout, out_px * 3));
    HIPCHK(hipMalloc(&dtiled, out_px * 3));
    HIPCHK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dimg, img.data(), img.size(), hipMemcpyHostToDevice));
    CHECK(p_nesr_forward(ctx, dx, 1, 3, H, W, dy, nullptr));
    CHECK(p_nesr_forward_u8(ctx, dimg, H, W, dout, 1, NESR_ROUND_NEAREST, nullptr));
    // the frame as RealESRGANer(tile=32, tile_pad=10) would evaluate it, bf16: this rank holds all rows (one rank)
    CHECK(p_nesr_forward_sharded_u8(ctx16, dimg, H, W, 32, 10, 0, dtiled, nullptr));
    HIPCHK(hipDeviceSynchronize());
    CHECK(p_nesr_check_status(ctx));
    CHECK(p_nesr_check_status(ctx16));
    std::vector<float> y(out_px * 3);
    std::vector<uint8_t> q(out_px * 3), qt(out_px * 3);
    HIPCHK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(q.data(), dout, q.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(qt.data(), dtiled, qt.size(), hipMemcpyDeviceToHost));
    int worst = 0, nonfinite = 0;
    for (size_t p = 0; p < out_px; ++p)
        for (int c = 0; c < 3; ++c) {
            const float v = y[(size_t)c * out_px + p];
            if (!std::isfinite(v)) { ++nonfinite; continue; }
            const int want = (int)std::nearbyint(std::fmin(std::fmax(v, 0.f), 1.f) * 255.0f);
            const int got = q[p * 3 + (2 - c)];
            worst = std::abs(want - got) > worst ? std::abs(want - got) : worst;
        }
    std::printf("output %dx%d, float vs fused-u8 entry: max difference %d LSB, non-finite %d\n", 2 * H, 2 * W, worst, nonfinite);
    // tiled bf16 against untiled f32: a different function of the input (tiles see 10 pixels of context, bf16 arithmetic) -- close, not equal
    double sad = 0;
    for (size_t i = 0; i < q.size(); ++i) sad += std::abs((int)q[i] - (int)qt[i]);
    std::printf("tiled bf16 (nesr_forward_sharded_u8, one rank) vs untiled f32: mean abs difference %.3f LSB\n", sad / q.size());
    int nt = 0, nm = 0;
    std::vector<int> t13(13 * 64), m4(4 * 64);
    CHECK(p_nesr_shard_plan(H, W, 2, 32, 10, 4, t13.data(), 64, &nt, m4.data(), 64, &nm));
    std::printf("plan for 4 ranks: %d tiles, owners", nt);
    for (int i = 0; i < nt; ++i) std::printf(" %d", t13[13 * i + 12]);
    std::printf("; %d row moves:", nm);
    for (int i = 0; i < nm; ++i) std::printf(" %d->%d[%d,%d)", m4[4 * i], m4[4 * i + 1], m4[4 * i + 2], m4[4 * i + 3]);
    std::printf("\n");
    p_nesr_destroy(ctx);
    p_nesr_destroy(ctx16);
    HIPCHK(hipFree(dx)); HIPCHK(hipFree(dy)); HIPCHK(hipFree(dimg)); HIPCHK(hipFree(dout)); HIPCHK(hipFree(dtiled));
    return (worst <= 0 && nonfinite == 0 && sad / q.size() < 8.0) ? 0 : 4;
}
