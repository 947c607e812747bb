// Micro-benchmark (diagnostic): the 3-piece bf16 split of an fp32 GEMM on v_mfma_f32_16x16x32_bf16, and how VALU work of
// another wave shares a SIMD with a wave that issues MFMAs back to back.
//   x = h + m + l, each piece the next 8 bits of the significand (truncation: exact);  a.b ~ hh + hm + mh + mm + hl + lh
//   (the terms ml, lm, ll, <= 2^-24 of |a||b| each, are dropped): 6 MFMAs of 4 passes per 32 k against 8 of 8 passes.
//   hipcc -O3 --offload-arch=gfx950 mfma_bf16x3_probe.hip -o mfma_bf16x3_probe && ./mfma_bf16x3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int K = 160;

__device__ __forceinline__ void split3(float x, short& h, short& m, short& l) {
    const unsigned b = __float_as_uint(x), hb = b & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(hb);                 // exact
    const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mb);                // exact, <= 8 significant bits
    h = (short)(hb >> 16);
    m = (short)(mb >> 16);
    l = (short)(__float_as_uint(r2) >> 16);
}

// A [16][K] row-major, B [K][16] row-major, C [16][16]; one wave.  MODE 0: fp32 MFMA, 1: 6 terms, 2: 4 terms (hh hm mh mm)
template <int MODE>
__global__ void tile_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C) {
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k + q], B[(k + q) * 16 + r], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 32) {
            bf16x8 ah, am, al, bh, bm, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                short h, m, l;
                split3(A[r * K + k0 + 8 * q + j], h, m, l); ah[j] = h; am[j] = m; al[j] = l;
                split3(B[(k0 + 8 * q + j) * 16 + r], h, m, l); bh[j] = h; bm[j] = m; bl[j] = l;
            }
            if (MODE == 1) {
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc2, 0, 0, 0);
            }
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
        }
        acc += acc2;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) C[(4 * q + e) * 16 + r] = acc[e];
}

// SIMD sharing: 512 threads = 2 waves per SIMD.  Waves 0..3 issue MFMAs back to back (MFMA_MODE 0: none, 1: fp32 16x16x4, 2:
// bf16 16x16x32), waves 4..7 run `valu_n` dependent-free fp32 FMAs (4 chains) and report their cycles.
template <int MFMA_MODE, int PRIO>
__global__ __launch_bounds__(512) void share_kernel(float* out, int mfma_n, int valu_n, unsigned long long* cyc_valu, unsigned long long* cyc_mfma) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_readcyclecounter();
    float res = 0.f;
    if (wave < 4) {
        f32x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float a = 1.f + lane * 1e-3f, b = 1.f - lane * 1e-3f;
        bf16x8 ah, bh;
        for (int j = 0; j < 8; ++j) { ah[j] = (short)(0x3F80 + lane + j); bh[j] = (short)(0x3F80 - j); }
        if (MFMA_MODE != 0)
            for (int it = 0; it < mfma_n; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (MFMA_MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
                }
            }
        f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
        res = s[0] + s[1] + s[2] + s[3];
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc_mfma[blockIdx.x] = t1 - t0;
    } else {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        float x0 = lane * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
        const float c = 1.0001f, d = 1e-4f;
        for (int it = 0; it < valu_n; ++it) {
            x0 = fmaf(x0, c, d); x1 = fmaf(x1, c, d); x2 = fmaf(x2, c, d); x3 = fmaf(x3, c, d);
        }
        res = x0 + x1 + x2 + x3;
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 256) cyc_valu[blockIdx.x] = t1 - t0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

int main() {
    std::vector<float> hA(16 * K), hB(K * 16);
    srand(1);
    for (auto& v : hA) v = (float)(rand() / (double)RAND_MAX) * 2.f - 1.f;
    for (auto& v : hB) v = ((float)(rand() / (double)RAND_MAX) * 2.f - 1.f) * 0.3f;
    std::vector<double> ref(256, 0.0);
    double scale = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)hA[i * K + k] * (double)hB[k * 16 + j];
            ref[i * 16 + j] = s;
            scale = fmax(scale, fabs(s));
        }
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, 256 * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    const char* names[3] = {"fp32 16x16x4 (40 MFMAs)", "bf16 3-piece split, 6 terms (30 MFMAs)", "bf16 3-piece split, 4 terms (20 MFMAs)"};
    std::vector<float> c0(256);
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(tile_kernel<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        if (mode == 1) hipLaunchKernelGGL(tile_kernel<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        if (mode == 2) hipLaunchKernelGGL(tile_kernel<2>, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        std::vector<float> hC(256);
        CK(hipMemcpy(hC.data(), dC, 256 * 4, hipMemcpyDeviceToHost));
        if (mode == 0) c0 = hC;
        double err = 0, dev = 0;
        for (int i = 0; i < 256; ++i) { err = fmax(err, fabs((double)hC[i] - ref[i])); dev = fmax(dev, fabs((double)hC[i] - (double)c0[i])); }
        printf("%-42s max |err| / max |C| = %.2e   max |C - C_fp32mfma| / max |C| = %.2e\n", names[mode], err / scale, dev / scale);
    }
    float* out;
    unsigned long long *cv, *cm;
    const int blocks = 256;
    CK(hipMalloc(&out, blocks * 512 * 4)); CK(hipMalloc(&cv, blocks * 8)); CK(hipMalloc(&cm, blocks * 8));
    const int mfma_n = 2000, valu_n = 4000;
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL((share_kernel<0, 0>), dim3(blocks), dim3(512), 0, 0, out, mfma_n, valu_n, cv, cm);
            if (mode == 1) hipLaunchKernelGGL((share_kernel<1, 0>), dim3(blocks), dim3(512), 0, 0, out, mfma_n, valu_n, cv, cm);
            if (mode == 2) hipLaunchKernelGGL((share_kernel<2, 0>), dim3(blocks), dim3(512), 0, 0, out, mfma_n, valu_n, cv, cm);
            if (mode == 3) hipLaunchKernelGGL((share_kernel<1, 1>), dim3(blocks), dim3(512), 0, 0, out, mfma_n, valu_n, cv, cm);
            if (mode == 4) hipLaunchKernelGGL((share_kernel<2, 1>), dim3(blocks), dim3(512), 0, 0, out, mfma_n, valu_n, cv, cm);
            CK(hipDeviceSynchronize());
        }
        unsigned long long hv[256], hm[256];
        CK(hipMemcpy(hv, cv, sizeof(hv), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hm, cm, sizeof(hm), hipMemcpyDeviceToHost));
        double a = 0, b = 0;
        for (int i = 0; i < blocks; ++i) { a += hv[i]; b += hm[i]; }
        printf("SIMD shared with %-22s: %d x 4 FMAs of the VALU wave take %.0f cycles; the MFMA wave's %d x 4 MFMAs %.0f cycles\n",
               mode == 0 ? "an idle wave" : (mode == 1 ? "fp32 16x16x4 MFMAs" : (mode == 2 ? "bf16 16x16x32 MFMAs" : (mode == 3 ? "fp32 MFMAs, VALU prio 3" : "bf16 MFMAs, VALU prio 3"))), valu_n, a / blocks, mfma_n, b / blocks);
    }
    return 0;
}
