// Micro-benchmarks of the fp32 MFMA building blocks used by shade.hip / shade_bwd.hip (diagnostic only).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off mfma_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// K1: register-only MFMA stream, NACC independent accumulators
template <int NACC>
__global__ __launch_bounds__(512) void k_pure(float* out, int iters) {
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        a += 1e-6f;
    }
    f32x4 s = acc[0];
    for (int j = 1; j < NACC; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// K2: the mma_block shape: A (weights) from global [F][ldw], B (activations) from LDS [64][ldx], 1 feature tile x 4
// sample tiles per wave, kgroups k-groups, repeated `reps` times (no barriers)
template <int UNROLL>
__global__ __launch_bounds__(512) void k_block(const float* __restrict__ W, int ldw, int kgroups, int reps, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ldx = kgroups * 16 + 4;
    for (int i = threadIdx.x; i < 64 * ldx; i += blockDim.x) lds[i] = (i % 17) * 0.01f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0, 0, 0, 0};
    const float* wp = W + (size_t)(16 * wave + r) * ldw + 4 * kq;
    const float* xp = lds + r * ldx + 4 * kq;
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll UNROLL
        for (int kg = 0; kg < kgroups; ++kg) {
            f32x4 a = *reinterpret_cast<const f32x4*>(wp + 16 * kg);
            f32x4 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4*>(xp + 16 * j * ldx + 16 * kg);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[j][e], acc[j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// K3: same but every A fragment of the phase is loaded up front (registers), then the MFMA loop touches LDS only
template <int KG>
__global__ __launch_bounds__(512) void k_preload(const float* __restrict__ W, int ldw, int reps, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ldx = KG * 16 + 4;
    for (int i = threadIdx.x; i < 64 * ldx; i += blockDim.x) lds[i] = (i % 17) * 0.01f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0, 0, 0, 0};
    const float* wp = W + (size_t)(16 * wave + r) * ldw + 4 * kq;
    const float* xp = lds + r * ldx + 4 * kq;
    for (int rep = 0; rep < reps; ++rep) {
        f32x4 a[KG];
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) a[kg] = *reinterpret_cast<const f32x4*>(wp + 16 * kg);
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            f32x4 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4*>(xp + 16 * j * ldx + 16 * kg);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kg][e], b[j][e], acc[j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// K4: the weight-gradient shape (mma_gen COL/COL): both operands from LDS with strided scalar reads, k = samples
template <int NBT>
__global__ __launch_bounds__(512) void k_col(int reps, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ld = 132;
    for (int i = threadIdx.x; i < 2 * 64 * ld; i += blockDim.x) lds[i] = (i % 17) * 0.01f;
    __syncthreads();
    const float* A = lds; const float* B = lds + 64 * ld;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
    f32x4 acc[NBT];
    for (int j = 0; j < NBT; ++j) acc[j] = (f32x4){0, 0, 0, 0};
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll 1
        for (int kg = 0; kg < 4; ++kg) {
            const int k = 16 * kg + 4 * kq;
            f32x4 a, b[NBT];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = A[(k + e) * ld + 16 * wave + r];
#pragma unroll
            for (int j = 0; j < NBT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) b[j][e] = B[(k + e) * ld + 16 * j + r];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NBT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[j][e], acc[j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    f32x4 s = acc[0];
    for (int j = 1; j < NBT; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <typename F>
float time_ms(F launch, int n = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / n;
}

int main() {
    float *out, *W;
    CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMalloc(&W, 128 * 160 * 4));
    std::vector<float> h(128 * 160, 0.01f);
    CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int nwg = 256;
    {
        const int iters = 4096;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_pure<4>, dim3(nwg), dim3(512), 0, 0, out, iters); });
        double fl = (double)nwg * 8 * iters * 16 * 2048.0;
        printf("pure MFMA 4 acc, 8 waves/CU : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_pure<1>, dim3(nwg), dim3(512), 0, 0, out, iters); });
        fl = (double)nwg * 8 * iters * 4 * 2048.0;
        printf("pure MFMA 1 acc (dependent)  : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_pure<4>, dim3(nwg), dim3(256), 0, 0, out, iters); });
        fl = (double)nwg * 4 * iters * 16 * 2048.0;
        printf("pure MFMA 4 acc, 4 waves/CU : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
    }
    for (int kgroups : {8, 10}) {
        const int reps = 512;
        const size_t lds = 64 * (kgroups * 16 + 4) * 4;
        double fl = (double)nwg * 8 * reps * kgroups * 16 * 2048.0;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_block<1>, dim3(nwg), dim3(512), lds, 0, W, 160, kgroups, reps, out); });
        printf("mma_block kg=%d unroll1       : %.3f ms  %.1f TFLOP/s\n", kgroups, ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_block<2>, dim3(nwg), dim3(512), lds, 0, W, 160, kgroups, reps, out); });
        printf("mma_block kg=%d unroll2       : %.3f ms  %.1f TFLOP/s\n", kgroups, ms, fl / ms / 1e9);
    }
    {
        const int reps = 512;
        double fl = (double)nwg * 8 * reps * 8 * 16 * 2048.0;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_preload<8>, dim3(nwg), dim3(512), 64 * 132 * 4, 0, W, 160, reps, out); });
        printf("preloaded A kg=8              : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
        fl = (double)nwg * 8 * reps * 10 * 16 * 2048.0;
        ms = time_ms([&] { hipLaunchKernelGGL(k_preload<10>, dim3(nwg), dim3(512), 64 * 164 * 4, 0, W, 160, reps, out); });
        printf("preloaded A kg=10             : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
    }
    {
        const int reps = 512;
        double fl = (double)nwg * 8 * reps * 4 * 4 * 8 * 2048.0;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_col<8>, dim3(nwg), dim3(512), 2 * 64 * 132 * 4, 0, reps, out); });
        printf("COL/COL 1x8 tiles             : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
        fl = (double)nwg * 8 * reps * 4 * 4 * 4 * 2048.0;
        ms = time_ms([&] { hipLaunchKernelGGL(k_col<4>, dim3(nwg), dim3(512), 2 * 64 * 132 * 4, 0, reps, out); });
        printf("COL/COL 1x4 tiles             : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms / 1e9);
    }
    return 0;
}
