#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* __restrict__ src, float* __restrict__ out, int n) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each wave DMAs 1 KiB: lane l -> lds[wave*256 + 4 l .. +3] from src[(wave*64 + (63 - l)) * 4] (reversed quads)
    const float* g = src + (size_t)(wave * 64 + (63 - lane)) * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = lds[i];
}
int main() {
    const int n = 1024;
    float *s, *o; hipMalloc(&s, n * 4); hipMalloc(&o, n * 4);
    float h[n]; for (int i = 0; i < n; ++i) h[i] = i;
    hipMemcpy(s, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), n * 4, 0, s, o, n);
    hipMemcpy(h, o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        float exp = (w * 64 + (63 - l)) * 4 + e;
        if (h[w * 256 + l * 4 + e] != exp) ++bad;
    }
    printf("bad=%d first=%g %g %g %g\n", bad, h[0], h[1], h[2], h[3]);
    return bad != 0;
}
