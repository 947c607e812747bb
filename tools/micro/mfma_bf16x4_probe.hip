// Micro-benchmark (diagnostic, for the next round's re-tiling of the shading kernels): one 16 x 16 output tile over K = 128
//   (a) exact fp32:  32 x v_mfma_f32_16x16x4_f32
//   (b) bf16 split:  a = a_hi + a_lo (hi = the upper 16 bits, lo = round-to-nearest bf16 of the remainder), four products
//                    hi.hi + hi.lo + lo.hi + lo.lo on v_mfma_f32_16x16x32_bf16 (16 MFMAs), fp32 accumulation
//   (c) the same without the lo.lo term (12 MFMAs)
// checked against a double-precision host product, then timed with the operands in registers.
//   hipcc -O3 --offload-arch=gfx950 mfma_bf16x4_probe.hip -o mfma_bf16x4_probe && ./mfma_bf16x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int K = 128;

__device__ __forceinline__ void split8(const float* x, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned b = __float_as_uint(x[j]);
        const float h = __uint_as_float(b & 0xFFFF0000u);
        const unsigned r = __float_as_uint(x[j] - h);                       // exact
        hi[j] = (short)(b >> 16);
        lo[j] = (short)((r + 0x7FFFu + ((r >> 16) & 1u)) >> 16);            // round to nearest even
    }
}

// A [16][K] row-major, B [K][16] row-major, C [16][16]; one wave
template <int MODE>
__global__ void tile_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C) {
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k + q], B[(k + q) * 16 + r], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 32) {
            float a[8], b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a[j] = A[r * K + k0 + 8 * q + j];
                b[j] = B[(k0 + 8 * q + j) * 16 + r];
            }
            bf16x8 ah, al, bh, bl;
            split8(a, ah, al);
            split8(b, bh, bl);
            if (MODE == 1) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) C[(4 * q + e) * 16 + r] = acc[e];          // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
}

// throughput: every wave of every CU repeats the K = 128 tile product on register operands (4 independent accumulators)
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = 1.f + lane * 1e-3f, b = 1.f - lane * 1e-3f;
    bf16x8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) { ah[j] = (short)(0x3F80 + lane + j); al[j] = (short)(0x3B00 + j); bh[j] = (short)(0x3F80 - j); bl[j] = (short)(0x3A80 + lane); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < K / 4; ++k) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < K / 32; ++k) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
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
    const char* names[3] = {"fp32 16x16x4 (32 MFMAs)", "bf16 split, 4 terms (16 MFMAs)", "bf16 split, 3 terms (12 MFMAs)"};
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(tile_kernel<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        if (mode == 1) hipLaunchKernelGGL(tile_kernel<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        if (mode == 2) hipLaunchKernelGGL(tile_kernel<2>, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        std::vector<float> hC(256);
        CK(hipMemcpy(hC.data(), dC, 256 * 4, hipMemcpyDeviceToHost));
        double err = 0;
        for (int i = 0; i < 256; ++i) err = fmax(err, fabs((double)hC[i] - ref[i]));
        printf("%-34s max |err| / max |C| = %.2e\n", names[mode], err / scale);
    }
    float* out;
    unsigned long long* cyc;
    const int blocks = 256, iters = 200;
    CK(hipMalloc(&out, blocks * 256 * 4)); CK(hipMalloc(&cyc, blocks * 8));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
            else hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
            CK(hipDeviceSynchronize());
        }
        unsigned long long h[256];
        CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        double c = 0;
        for (int i = 0; i < blocks; ++i) c += h[i];
        c /= blocks;
        printf("%-34s %.0f shader cycles per 16x16x128 tile product and wave (one wave per SIMD, operands in registers)\n",
               mode == 0 ? "fp32 16x16x4" : "bf16 split, 4 terms", c / (iters * 4.0));
    }
    return 0;
}
