// Micro-benchmark (diagnostic only): cost of LDS float atomics (ds_add_f32, no return) against a plain
// read-add-write and a plain read, conflict-free lane -> address maps, per wave-instruction and per CU.
//   hipcc -O3 --offload-arch=gfx950 lds_atomic_probe.hip -o lds_atomic_probe && ./lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// MODE 0: ds_add_f32 (atomicAdd, result unused)   1: plain read + add + write   2: read only (sum in a register)
// 3: ds_add_f32 with only lanes (lane & 3) < 2 active (the line-gradient shape)   4: atomics, 4-lane groups on ONE address
template <int MODE>
__global__ __launch_bounds__(512) void k_lds(float* out, int iters, int nrows, unsigned long long* cyc) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < nrows * 64; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f, v = 1.0f + lane * 1e-3f;
    unsigned r = wave * 7 + 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            r = r * 1664525u + 1013904223u;
            const int row = (r >> 8) % nrows;              // wave-uniform pseudo-random row of 64 floats
            float* p = lds + row * 64 + lane;
            if (MODE == 0) atomicAdd(p, v);
            if (MODE == 1) *p = *p + v;
            if (MODE == 2) acc += *p;
            if (MODE == 3) { if ((lane & 3) < 2) atomicAdd(p, v); }
            if (MODE == 4) atomicAdd(lds + row * 64 + (lane >> 2), v);
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = acc;
    for (int i = threadIdx.x; i < nrows * 64; i += blockDim.x) s += lds[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, int waves) {
    const int blocks = 256, iters = 2000, nrows = 128;
    float* out;
    unsigned long long* cyc;
    CK(hipMalloc(&out, blocks * 512 * 4));
    CK(hipMalloc(&cyc, blocks * 8));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_lds<MODE>, dim3(blocks), dim3(64 * waves), nrows * 64 * 4, 0, out, iters, nrows, cyc);
        hipEventRecord(b);
        CK(hipDeviceSynchronize());
    }
    float ms;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long h[256];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double c = 0;
    for (int i = 0; i < blocks; ++i) c += h[i];
    c /= blocks;
    const double ops = (double)iters * 8 * waves;           // wave-instructions per CU
    printf("%-28s waves/CU %2d: %.3f ms, %.1f shader cycles per wave-op per CU (%.1f per wave)\n", name, waves, ms, c / ops, c / (iters * 8.0));
    hipFree(out);
    hipFree(cyc);
    return 0;
}

int main() {
    for (int w : {1, 4, 8}) {
        if (run<0>("ds_add_f32", w)) return 1;
        if (run<1>("plain read-add-write", w)) return 1;
        if (run<2>("read only", w)) return 1;
        if (run<3>("ds_add_f32 half lanes", w)) return 1;
        if (run<4>("ds_add_f32 4 lanes/address", w)) return 1;
    }
    return 0;
}
