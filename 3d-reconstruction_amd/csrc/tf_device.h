// tf_device.h — device-side helpers shared by the gfx950 kernels.
//
// Numerics contract: this translation unit is compiled with -ffp-contract=off so that
// every `a*b+c` written below is an UNFUSED multiply-then-add exactly like the eager PyTorch ops it
// restates (sample positions and the bbox / alpha-mask predicates must be bit-exact).  Where a fused
// multiply-add is wanted (feature accumulation), it is written explicitly as fmaf().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tensorf_hip.h"

#define TF_CHECK_LAUNCH() (int)hipGetLastError()

namespace tf {

typedef float float4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---- cross-lane helpers (wave64) -------------------------------------------------------------
// quad_perm DPP: exchange within groups of 4 lanes without touching LDS.
__device__ __forceinline__ float quad_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));  // [1,0,3,2]
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));  // [2,3,0,1]
}
__device__ __forceinline__ float quad_sum(float v) {
    v += quad_xor1(v);
    v += quad_xor2(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// number of set bits of `m` strictly below this lane
__device__ __forceinline__ int prefix_popc(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}
// inclusive product scan across the 64 lanes
__device__ __forceinline__ float wave_scan_mul(float v) {
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float up = __shfl_up(v, o, 64);
        if (lane >= o) v *= up;
    }
    return v;
}
// inclusive sum scan across the 64 lanes
__device__ __forceinline__ float wave_scan_add(float v) {
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float up = __shfl_up(v, o, 64);
        if (lane >= o) v += up;
    }
    return v;
}

// XCD-aware block remap (bijective for any grid size): blocks b and b+8 share an XCD's L2, so give each
// XCD a contiguous range of work items.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// ---- ray geometry -------------------------------------------------------------------------------
struct Ray {
    float o[3], d[3];
    float tmin;   // AABB mode
    float jit;    // AABB mode: stratified jitter (0 in eval)
    float dnorm;  // NDC mode: |d|
};

// tensorBase.py:193-196
__device__ __forceinline__ float slab_tmin(const TfField& F, const Ray& r) {
    float t = -INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float v = (r.d[a] == 0.f) ? 1e-6f : r.d[a];
        float ta = (F.aabb_hi[a] - r.o[a]) / v;
        float tb = (F.aabb_lo[a] - r.o[a]) / v;
        t = fmaxf(t, fminf(ta, tb));
    }
    return fminf(fmaxf(t, F.near_), F.far_);
}

// z of sample i.  AABB mode: t_min + stepSize*(i + jitter)   (tensorBase.py:198-203)
//                 NDC  mode: shared table (linspace [+ jitter]) (tensorBase.py:181-183)
__device__ __forceinline__ float sample_z(const TfField& F, const Ray& r, const float* ztab, int i) {
    if (ztab) return ztab[i];
    float rng = (float)i + r.jit;
    float st = F.step * rng;
    return r.tmin + st;
}

__device__ __forceinline__ void sample_pos(const Ray& r, float z, float p[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float m = r.d[a] * z;
        p[a] = r.o[a] + m;
    }
}

// tensorBase.py:206 — strict compares: points exactly on a face are inside
__device__ __forceinline__ bool in_bbox(const TfField& F, const float p[3]) {
    bool out = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) out |= (F.aabb_lo[a] > p[a]) | (p[a] > F.aabb_hi[a]);
    return !out;
}

// tensorBase.py:130-131
__device__ __forceinline__ void normalize(const TfField& F, const float p[3], float u[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float t = p[a] - F.aabb_lo[a];
        float s = t * F.inv_aabb[a];
        u[a] = s - 1.f;
    }
}

// grid_sample unnormalize, align_corners=True
__device__ __forceinline__ float unnorm(float u, int size) { return ((u + 1.f) * 0.5f) * (float)(size - 1); }

// AlphaGridMask.sample_alpha(p) > 0   (tensorBase.py:41-48, 350-351) through the 1-byte cell table.
__device__ __forceinline__ bool alpha_hit(const TfField& F, const float p[3]) {
    float f[3];
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float t = p[a] - F.alpha_lo[a];
        float s = t * F.alpha_inv[a];
        float u = s - 1.f;
        float x = unnorm(u, F.alpha_grid[a]);
        if (!(x > -1.f && x < (float)F.alpha_grid[a])) return false;  // every corner out of bounds (or NaN)
        float x0 = floorf(x);
        f[a] = x - x0;
        c[a] = (int)x0 + 1;  // 0 .. G
    }
    const int sx = F.alpha_grid[0] + 1, sy = F.alpha_grid[1] + 1;
    uint32_t m = F.alpha_cells[((size_t)c[2] * sy + c[1]) * sx + c[0]];
    uint32_t allow = (f[0] > 0.f ? 0xFFu : 0x55u) & (f[1] > 0.f ? 0xFFu : 0x33u) & (f[2] > 0.f ? 0xFFu : 0x0Fu);
    return (m & allow) != 0;
}

// ---- bilinear / linear taps on channel-last factor tensors ---------------------------------------
struct Tap2 {   // 4 texel offsets (in texels, multiply by C) + weights; out-of-range taps get weight 0
    int o00, o01, o10, o11;
    float w00, w01, w10, w11;
};
struct Tap1 {
    int o0, o1;
    float w0, w1;
};

__device__ __forceinline__ void tap_axis(float u, int size, int& i0, int& i1, float& w0, float& w1) {
    float x = unnorm(u, size);
    x = fminf(fmaxf(x, -2.f), (float)size + 1.f);   // keeps the int conversion defined; NaN -> -2 (all zero)
    float x0 = floorf(x);
    float f = x - x0;
    float e = 1.f - f;
    int a = (int)x0, b = a + 1;
    bool va = (a >= 0) & (a < size), vb = (b >= 0) & (b < size);
    w0 = va ? e : 0.f;
    w1 = vb ? f : 0.f;
    i0 = va ? a : 0;
    i1 = vb ? b : 0;
}

__device__ __forceinline__ Tap2 make_tap2(float ux, float uy, int W, int H) {
    int x0, x1, y0, y1;
    float ex, fx, ey, fy;
    tap_axis(ux, W, x0, x1, ex, fx);
    tap_axis(uy, H, y0, y1, ey, fy);
    Tap2 t;
    t.o00 = y0 * W + x0; t.o01 = y0 * W + x1; t.o10 = y1 * W + x0; t.o11 = y1 * W + x1;
    t.w00 = ey * ex; t.w01 = ey * fx; t.w10 = fy * ex; t.w11 = fy * fx;
    return t;
}
__device__ __forceinline__ Tap1 make_tap1(float u, int G) {
    Tap1 t;
    tap_axis(u, G, t.o0, t.o1, t.w0, t.w1);
    return t;
}

__device__ __forceinline__ float4_t ld4(const float* p) { return *reinterpret_cast<const float4_t*>(p); }

__device__ __forceinline__ float4_t bilerp4(const float* base, int C, const Tap2& t, int ch) {
    float4_t a = ld4(base + (size_t)t.o00 * C + ch);
    float4_t b = ld4(base + (size_t)t.o01 * C + ch);
    float4_t c = ld4(base + (size_t)t.o10 * C + ch);
    float4_t d = ld4(base + (size_t)t.o11 * C + ch);
    float4_t r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = fmaf(d[k], t.w11, fmaf(c[k], t.w10, fmaf(b[k], t.w01, a[k] * t.w00)));
    return r;
}
__device__ __forceinline__ float4_t lerp4(const float* base, int C, const Tap1& t, int ch) {
    float4_t a = ld4(base + (size_t)t.o0 * C + ch);
    float4_t b = ld4(base + (size_t)t.o1 * C + ch);
    float4_t r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = fmaf(b[k], t.w1, a[k] * t.w0);
    return r;
}
__device__ __forceinline__ float bilerp1(const float* base, int C, const Tap2& t, int ch) {
    return fmaf(base[(size_t)t.o11 * C + ch], t.w11,
                fmaf(base[(size_t)t.o10 * C + ch], t.w10,
                     fmaf(base[(size_t)t.o01 * C + ch], t.w01, base[(size_t)t.o00 * C + ch] * t.w00)));
}
__device__ __forceinline__ float lerp1(const float* base, int C, const Tap1& t, int ch) {
    return fmaf(base[(size_t)t.o1 * C + ch], t.w1, base[(size_t)t.o0 * C + ch] * t.w0);
}

// index maps of the VM decomposition (tensorBase.py:60-61): plane i spans (matMode[i][0] -> x/W,
// matMode[i][1] -> y/H); line i runs along vecMode[i].
__device__ __forceinline__ int mat0(int i) { return i == 2 ? 1 : 0; }
__device__ __forceinline__ int mat1(int i) { return i == 0 ? 1 : 2; }
__device__ __forceinline__ int vecm(int i) { return 2 - i; }

struct VmTaps {
    Tap2 p[3];
    Tap1 l[3];
};
__device__ __forceinline__ void make_vm_taps(const int grid[3], const float u[3], VmTaps& t) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        t.p[i] = make_tap2(u[mat0(i)], u[mat1(i)], grid[mat0(i)], grid[mat1(i)]);
        t.l[i] = make_tap1(u[vecm(i)], grid[vecm(i)]);
    }
}

// Partial density feature of one sample computed by ONE lane of a 4-lane group: the lane with
// sub-index `sub` covers channel quads sub, sub+4, ... so that the four lanes of a group read one
// contiguous 64-B segment per tap.  Caller reduces with quad_sum().
// VM: sum_i sum_c (P*m)(L*m)  tensoRF.py:215-225;  CP: sum_c (L0*L1*L2)*m  tensoRF.py:363-384.
__device__ __forceinline__ float density_partial(int model, const TfFactors& D, const int grid[3], const float u[3],
                                                 int sub) {
    float acc = 0.f;
    if (model == TF_MODEL_VM) {
        VmTaps t;
        make_vm_taps(grid, u, t);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int C = D.n_comp[i];
            const float* mk = D.mask[i];
            if ((C & 3) == 0) {
                for (int q = sub; q < (C >> 2); q += 4) {
                    float4_t p = bilerp4(D.plane[i], C, t.p[i], q * 4);
                    float4_t l = lerp4(D.line[i], C, t.l[i], q * 4);
                    if (mk) {
                        float4_t m = ld4(mk + q * 4);
                        p *= m;
                        l *= m;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc = fmaf(p[k], l[k], acc);
                }
            } else {
                for (int c = sub; c < C; c += 4) {
                    float p = bilerp1(D.plane[i], C, t.p[i], c);
                    float l = lerp1(D.line[i], C, t.l[i], c);
                    if (mk) {
                        p *= mk[c];
                        l *= mk[c];
                    }
                    acc = fmaf(p, l, acc);
                }
            }
        }
    } else {
        const int C = D.n_comp[0];
        Tap1 t0 = make_tap1(u[vecm(0)], grid[vecm(0)]);
        Tap1 t1 = make_tap1(u[vecm(1)], grid[vecm(1)]);
        Tap1 t2 = make_tap1(u[vecm(2)], grid[vecm(2)]);
        const float* mk = D.mask[0];
        if ((C & 3) == 0) {
            for (int q = sub; q < (C >> 2); q += 4) {
                float4_t v = lerp4(D.line[0], C, t0, q * 4);
                v *= lerp4(D.line[1], C, t1, q * 4);
                v *= lerp4(D.line[2], C, t2, q * 4);
                if (mk) v *= ld4(mk + q * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc += v[k];
            }
        } else {
            for (int c = sub; c < C; c += 4) {
                float v = lerp1(D.line[0], C, t0, c) * lerp1(D.line[1], C, t1, c);
                v *= lerp1(D.line[2], C, t2, c);
                if (mk) v *= mk[c];
                acc += v;
            }
        }
    }
    return acc;
}

// feature2density (tensorBase.py:291-295); F.softplus = x > 20 ? x : log1p(exp(x))
__device__ __forceinline__ float density_act(const TfField& F, float f) {
    if (F.act == TF_ACT_RELU) return fmaxf(f, 0.f);
    float x = f + F.density_shift;
    return x > 20.f ? x : log1pf(expf(x));
}
// d sigma / d f
__device__ __forceinline__ float density_act_grad(const TfField& F, float f) {
    if (F.act == TF_ACT_RELU) return f > 0.f ? 1.f : 0.f;
    float x = f + F.density_shift;
    return x > 20.f ? 1.f : 1.f / (1.f + expf(-x));
}

}  // namespace tf
