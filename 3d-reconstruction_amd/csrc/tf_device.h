// tf_device.h — device-side helpers shared by the gfx950 kernels.
//
// Numerics contract: this translation unit is compiled with -ffp-contract=off so that
// every `a*b+c` written below is an UNFUSED multiply-then-add exactly like the eager PyTorch ops it
// restates (sample positions and the bbox / alpha-mask predicates must be bit-exact).  Where a fused
// multiply-add is wanted (feature accumulation), it is written explicitly as fmaf().
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include "../../include/tensorf_hip.h"

#define TF_CHECK_LAUNCH() (int)hipGetLastError()

namespace tf {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and size: the attribute is a ceiling, so a launch needs
// the call only when it asks for more than any launch before it (the eager training step made 11 of these calls per step,
// each a few microseconds of host time on a host-bound path).  Host code runs on the caller's single thread.
inline hipError_t ensure_dynamic_lds(const void* fn, size_t bytes) {
    static const void* seen_fn[64];
    static size_t seen_bytes[64];
    static int n_seen = 0;
    for (int i = 0; i < n_seen; ++i)
        if (seen_fn[i] == fn) {
            if (bytes <= seen_bytes[i]) return hipSuccess;
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) seen_bytes[i] = bytes;
            return e;
        }
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && n_seen < 64) {
        seen_fn[n_seen] = fn;
        seen_bytes[n_seen++] = bytes;
    }
    return e;
}

// Diagnostic build only (-DTF_PHASE_TIMING): switches to ablate the atomic traffic (bit 0: skip plane
// atomics, bit 1: skip line atomics).  Compiled out of the shipped library.
#ifdef TF_PHASE_TIMING
static __device__ int tf_dbg_flags;
#define TF_SKIP_PLANE_ATOMICS (tf_dbg_flags & 1)
#define TF_SKIP_LINE_ATOMICS (tf_dbg_flags & 2)
#else
#define TF_SKIP_PLANE_ATOMICS 0
#define TF_SKIP_LINE_ATOMICS 0
#endif

// Diagnostic build only (-DTF_PHASE_TIMING, lib/libtensorf_hip_diag.so): per-phase shader-clock totals.
#ifdef TF_PHASE_TIMING
static __device__ unsigned long long tf_phase_cycles[16];
#define TF_T0() unsigned long long _t = __builtin_readcyclecounter(); unsigned long long _ph[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}
#define TF_MARK(i) do { unsigned long long _n = __builtin_readcyclecounter(); _ph[i] += _n - _t; _t = _n; } while (0)
static __device__ unsigned long long tf_phase_cycles_w4[16];      // the same, seen by wave 4 (thread 256)
#define TF_FLUSH() do { if (threadIdx.x == 0) for (int _i = 0; _i < 16; ++_i) atomicAdd(&tf_phase_cycles[_i], _ph[_i]); \
                        if (threadIdx.x == 256) for (int _i = 0; _i < 16; ++_i) atomicAdd(&tf_phase_cycles_w4[_i], _ph[_i]); } while (0)
#else
#define TF_T0()
#define TF_MARK(i)
#define TF_FLUSH()
#endif

typedef float float4_t __attribute__((ext_vector_type(4)));

// One workgroup's share of a TfPackJob: row `by` of a (gx, n + 1) grid of 256-thread workgroups — rows 0 .. n-1 write the
// padded (or transposed) weight copies, row n zeroes the job's counter / histogram block.  Shared by tf_pack_matrices and
// the staging launch of the captured training step (tf_gather_batch_staged).
__device__ __forceinline__ void pack_block(const TfPackJob& J, int bx, int by, int gx) {
    if (by == J.n) {
        for (int i = bx * 256 + (int)threadIdx.x; i < J.n_zero; i += gx * 256) J.zero[i] = 0;
        return;
    }
    const TfPackItem& P = J.item[by];
    const int kp = (P.cols + 15) & ~15, total = P.rows_pad * kp;
    for (int i = bx * 256 + (int)threadIdx.x; i < total; i += gx * 256) {
        int r, c;
        if (P.transpose) {
            c = i / P.rows_pad;
            r = i - c * P.rows_pad;
        } else {
            r = i / kp;
            c = i - r * kp;
        }
        P.dst[i] = (r < P.rows && c < P.cols) ? P.src[(size_t)r * P.cols + c] : 0.f;
    }
}
__host__ inline int pack_grid_x(const TfPackJob& J) {      // workgroups per row: the largest item sets it
    int most = 0;
    for (int k = 0; k < J.n; ++k) {
        const int total = J.item[k].rows_pad * ((J.item[k].cols + 15) & ~15);
        most = total > most ? total : most;
    }
    return (most + 255) / 256;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains this wave's outstanding GLOBAL
// loads and stores (s_waitcnt vmcnt(0)), which stalls kernels that keep global stores (or prefetches of the next
// operands) in flight across phase boundaries.  Use only where the data exchanged between waves lives in LDS.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), result in every lane of the row; VALU only (DPP)
__device__ __forceinline__ float row16_sum(float v) {
    v = quad_sum(v);
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x124, 0xF, 0xF, true));   // row_ror:4
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, true));   // row_ror:8
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

// alpha_hit in two steps, so that a caller can have several cell loads in flight: the cell's index and the corner bits that
// count for this point (false: every corner out of bounds — no hit, nothing to load) ...
__device__ __forceinline__ bool alpha_cell(const TfField& F, const float p[3], size_t& cell, uint32_t& allow) {
    float f[3];
    int c[3];
    bool in = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float t = p[a] - F.alpha_lo[a];
        float s = t * F.alpha_inv[a];
        float u = s - 1.f;
        float x = unnorm(u, F.alpha_grid[a]);
        in &= (x > -1.f) & (x < (float)F.alpha_grid[a]);      // (NaN: out)
        float x0 = floorf(x);
        f[a] = x - x0;
        c[a] = (int)x0 + 1;  // 0 .. G
    }
    const int sx = F.alpha_grid[0] + 1, sy = F.alpha_grid[1] + 1;
    cell = in ? ((size_t)c[2] * sy + c[1]) * sx + c[0] : 0;
    allow = (f[0] > 0.f ? 0xFFu : 0x55u) & (f[1] > 0.f ? 0xFFu : 0x33u) & (f[2] > 0.f ? 0xFFu : 0x0Fu);
    return in;
}
// ... and the test of the loaded byte: (F.alpha_cells[cell] & allow) != 0

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

// ---- gradient scatter (backward of the VM / CP lookups) ---------------------------------------------
// One whole wave scatters the factor gradients of up to TWO samples (A, B).  `dprod(s, i, ch)` returns
// dL/d(product) of sample s (0 = A, 1 = B) for component ch of plane/line pair i; for the density field
// that is dL/df, for the appearance field dL/dV[coff_i + ch].
//
// Lane -> (tap, channel) maps follow MEMORY order, so that each atomic wave-instruction covers 256
// contiguous bytes or two 128-B segments in two rows (MI355X_MICROARCH 'Global float atomics'):
//   plane i : footprint of one sample = 2 rows x (2 taps x C floats contiguous); j in [0,4C) -> row, tap, c
//   line  i : footprint of one sample = 2 taps x C floats contiguous; two samples -> j in [0,4C)
// The work is split into a GATHER step (all loads of a pair, values + destination addresses into
// registers) and a COMMIT step (the atomics, back to back), so callers can issue the gather of the next
// pair before committing the current one: loads never wait behind the atomics they do not depend on.
// The line tensors are scattered into replica (blockIdx % n_rep).
struct PairVals {
    float v[9];    // [3i+0] plane i / sample A, [3i+1] plane i / sample B, [3i+2] line i (sample chosen per lane)
    float* a[9];
};

template <class DP>
__device__ __forceinline__ void vm_pair_gather(const TfFactors& F, const TfFactorGrads& G, const int grid[3],
                                               const float uA[3], const float uB[3], bool hasA, bool hasB, DP dprod,
                                               int lane, int it, PairVals& out) {
    const size_t rep = (size_t)(blockIdx.x % G.n_rep) * G.rep_stride;
    const int j = lane + 64 * it;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int C = F.n_comp[i], C2 = 2 * C, C4 = 4 * C;
        const float* mk = F.mask[i];
        out.v[3 * i] = out.v[3 * i + 1] = out.v[3 * i + 2] = 0.f;
        out.a[3 * i] = out.a[3 * i + 1] = G.plane[i];
        out.a[3 * i + 2] = G.line[i];
        if (j >= C4) continue;
        const int hi = j >= C2, o = j - (hi ? C2 : 0), tx = o >= C, c = o - (tx ? C : 0);
        const float m2 = mk ? mk[c] * mk[c] : 1.f;                 // (P m)(L m): the mask enters squared
        const Tap2 tpA = make_tap2(uA[mat0(i)], uA[mat1(i)], grid[mat0(i)], grid[mat1(i)]);
        const Tap2 tpB = make_tap2(uB[mat0(i)], uB[mat1(i)], grid[mat0(i)], grid[mat1(i)]);
        const Tap1 tlA = make_tap1(uA[vecm(i)], grid[vecm(i)]);
        const Tap1 tlB = make_tap1(uB[vecm(i)], grid[vecm(i)]);
        // plane gradient: hi = row of the 2x2 footprint
        if (hasA) {
            const int off = hi ? (tx ? tpA.o11 : tpA.o10) : (tx ? tpA.o01 : tpA.o00);
            const float w = hi ? (tx ? tpA.w11 : tpA.w10) : (tx ? tpA.w01 : tpA.w00);
            out.v[3 * i] = lerp1(F.line[i], C, tlA, c) * dprod(0, i, c) * (w * m2);
            out.a[3 * i] = G.plane[i] + (size_t)off * C + c;
        }
        if (hasB) {
            const int off = hi ? (tx ? tpB.o11 : tpB.o10) : (tx ? tpB.o01 : tpB.o00);
            const float w = hi ? (tx ? tpB.w11 : tpB.w10) : (tx ? tpB.w01 : tpB.w00);
            out.v[3 * i + 1] = lerp1(F.line[i], C, tlB, c) * dprod(1, i, c) * (w * m2);
            out.a[3 * i + 1] = G.plane[i] + (size_t)off * C + c;
        }
        // line gradient: hi = which sample
        if (hi ? hasB : hasA) {
            const Tap2& tp = hi ? tpB : tpA;
            const Tap1& tl = hi ? tlB : tlA;
            out.v[3 * i + 2] = bilerp1(F.plane[i], C, tp, c) * dprod(hi, i, c) * ((tx ? tl.w1 : tl.w0) * m2);
            out.a[3 * i + 2] = G.line[i] + rep + (size_t)(tx ? tl.o1 : tl.o0) * C + c;
        }
    }
}

// CP: f = sum_c L0 L1 L2 m; d/dL_i = prod_{j != i} L_j * m   (tensoRF.py:363-384, 394-415).  Only v[3i+2] is used.
template <class DP>
__device__ __forceinline__ void cp_pair_gather(const TfFactors& F, const TfFactorGrads& G, const int grid[3],
                                               const float uA[3], const float uB[3], bool hasA, bool hasB, DP dprod,
                                               int lane, int it, PairVals& out) {
    const size_t rep = (size_t)(blockIdx.x % G.n_rep) * G.rep_stride;
    const int C = F.n_comp[0], C2 = 2 * C, C4 = 4 * C;
    const int j = lane + 64 * it;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        out.v[k] = 0.f;
        out.a[k] = G.line[k / 3];
    }
    if (j >= C4) return;
    const int hi = j >= C2, o = j - (hi ? C2 : 0), tx = o >= C, c = o - (tx ? C : 0);
    if (!(hi ? hasB : hasA)) return;
    const float* u = hi ? uB : uA;
    Tap1 t[3];
    float l[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        t[i] = make_tap1(u[vecm(i)], grid[vecm(i)]);
        l[i] = lerp1(F.line[i], C, t[i], c);
    }
    const float d = dprod(hi, 0, c) * (F.mask[0] ? F.mask[0][c] : 1.f);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float others = i == 0 ? l[1] * l[2] : (i == 1 ? l[0] * l[2] : l[0] * l[1]);
        out.v[3 * i + 2] = d * others * (tx ? t[i].w1 : t[i].w0);
        out.a[3 * i + 2] = G.line[i] + rep + (size_t)(tx ? t[i].o1 : t[i].o0) * C + c;
    }
}

__device__ __forceinline__ void pair_commit(const PairVals& p) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if ((k % 3 == 2) ? TF_SKIP_LINE_ATOMICS : TF_SKIP_PLANE_ATOMICS) continue;
        if (p.v[k] != 0.f) atomicAdd(p.a[k], p.v[k]);
    }
}

// ---- run-length merged scatter (VM) ------------------------------------------------------------------
// Consecutive samples of a ray are half a voxel apart, so they mostly share their 2x2 plane footprint and
// their 2-tap line footprint.  The merged scatter therefore works on a CHUNK of up to 16 consecutive samples
// of one wave in two steps:
//   gather : 4 lanes per sample read the plane / line values (16-B pieces, like the forward) and park
//            P m^2 and L m^2 plus the per-sample tap geometry in LDS;
//   merge  : lanes are laid out in footprint MEMORY order (row, tap, channel); the wave walks the chunk's
//            samples in order, accumulating in a register while the footprint base stays the same and
//            issuing ONE atomic per lane when it changes (or the chunk ends).
// On the benchmark scene this removes ~60 % of the plane atomics and ~75 % of the line atomics.
struct TapMeta {      // geometry of one sample on plane/line pair i
    int x0, y0;       // footprint base texel (bilinear floor)
    float fx, fy;     // fractions; weights are (1-f, f) per axis
    int l0;           // line base entry
    float lf;
};
constexpr int kChunk = 16;

// words of LDS one wave needs for a chunk with `ctot` components in total
__host__ __device__ inline int chunk_lds_words(int ctot) { return kChunk * 3 * 6 + 2 * kChunk * ctot; }

__device__ __forceinline__ void tap_floor(float u, int size, int& i0, float& f) {
    float x = unnorm(u, size);
    x = fminf(fmaxf(x, -2.f), (float)size + 1.f);
    const float x0 = floorf(x);
    f = x - x0;
    i0 = (int)x0;
}

// Gather step.  `u_of(s, u)` fills the normalised coordinate of chunk sample s.  Layout in `buf`:
// TapMeta meta[kChunk][3]; float Ps[kChunk][ctot]; float Ls[kChunk][ctot].
template <class UF>
__device__ __forceinline__ void vm_chunk_gather(const TfFactors& F, const int grid[3], int ctot, int ns, UF u_of,
                                                float* buf, int lane) {
    TapMeta* meta = reinterpret_cast<TapMeta*>(buf);
    float* Ps = buf + kChunk * 3 * 6;
    float* Ls = Ps + kChunk * ctot;
    const int s = lane >> 2, sub = lane & 3;
    if (s >= ns) return;
    float u[3];
    u_of(s, u);
    VmTaps t;
    make_vm_taps(grid, u, t);
    int coff = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int C = F.n_comp[i];
        const float* mk = F.mask[i];
        if (sub == 0) {
            TapMeta m;
            tap_floor(u[mat0(i)], grid[mat0(i)], m.x0, m.fx);
            tap_floor(u[mat1(i)], grid[mat1(i)], m.y0, m.fy);
            tap_floor(u[vecm(i)], grid[vecm(i)], m.l0, m.lf);
            meta[s * 3 + i] = m;
        }
        float* ps = Ps + s * ctot + coff;
        float* ls = Ls + s * ctot + coff;
        if ((C & 3) == 0 && (coff & 3) == 0 && (ctot & 3) == 0) {
            for (int q = sub; q < (C >> 2); q += 4) {
                float4_t p = bilerp4(F.plane[i], C, t.p[i], q * 4);
                float4_t l = lerp4(F.line[i], C, t.l[i], q * 4);
                if (mk) {
                    const float4_t m = ld4(mk + q * 4);
                    p *= m * m;                                  // (P m)(L m): the mask enters squared
                    l *= m * m;
                }
                *reinterpret_cast<float4_t*>(ps + q * 4) = p;
                *reinterpret_cast<float4_t*>(ls + q * 4) = l;
            }
        } else {
            for (int c = sub; c < C; c += 4) {
                const float m2 = mk ? mk[c] * mk[c] : 1.f;
                ps[c] = bilerp1(F.plane[i], C, t.p[i], c) * m2;
                ls[c] = lerp1(F.line[i], C, t.l[i], c) * m2;
            }
        }
        coff += C;
    }
}

// Merge + scatter step.  `dprod(s, coff_i + c)` = dL/d(product) of chunk sample s, component coff_i + c.
template <class DP>
__device__ __forceinline__ void vm_chunk_merge_scatter(const TfFactors& F, const TfFactorGrads& G, const int grid[3],
                                                       int ctot, int ns, DP dprod, const float* buf, int lane) {
    const TapMeta* meta = reinterpret_cast<const TapMeta*>(buf);
    const float* Ps = buf + kChunk * 3 * 6;
    const float* Ls = Ps + kChunk * ctot;
    const size_t rep = (size_t)(blockIdx.x % G.n_rep) * G.rep_stride;
    int coff = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int C = F.n_comp[i], W = grid[mat0(i)], Hh = grid[mat1(i)], Gl = grid[vecm(i)];
        // ---------- plane gradient: j -> (row, tap, channel) in memory order
        if (!TF_SKIP_PLANE_ATOMICS) {
            for (int j0 = 0; j0 < 4 * C; j0 += 64) {
                const int j = j0 + lane;
                const bool on = j < 4 * C;
                const int row = j >= 2 * C, o = j - (row ? 2 * C : 0), tx = o >= C, c = o - (tx ? C : 0);
                float acc = 0.f;
                int kx = INT_MIN, ky = INT_MIN;
                float* gp = G.plane[i];
                for (int s = 0; s < ns; ++s) {
                    const TapMeta& m = meta[s * 3 + i];
                    if (m.x0 != kx || m.y0 != ky) {                    // wave-uniform: new footprint
                        if (on && acc != 0.f) atomicAdd(gp + ((size_t)(ky + row) * W + (kx + tx)) * C + c, acc);
                        acc = 0.f;
                        kx = m.x0;
                        ky = m.y0;
                    }
                    const bool ok = on && (unsigned)(kx + tx) < (unsigned)W && (unsigned)(ky + row) < (unsigned)Hh;
                    if (ok) {
                        const float w = (row ? m.fy : 1.f - m.fy) * (tx ? m.fx : 1.f - m.fx);
                        acc = fmaf(dprod(s, coff + c) * Ls[s * ctot + coff + c], w, acc);
                    }
                }
                if (on && acc != 0.f) atomicAdd(gp + ((size_t)(ky + row) * W + (kx + tx)) * C + c, acc);
            }
        }
        // ---------- line gradient: j -> (tap, channel)
        if (!TF_SKIP_LINE_ATOMICS) {
            for (int j0 = 0; j0 < 2 * C; j0 += 64) {
                const int j = j0 + lane;
                const bool on = j < 2 * C;
                const int tx = j >= C, c = j - (tx ? C : 0);
                float acc = 0.f;
                int kl = INT_MIN;
                float* gl = G.line[i] + rep;
                for (int s = 0; s < ns; ++s) {
                    const TapMeta& m = meta[s * 3 + i];
                    if (m.l0 != kl) {
                        if (on && acc != 0.f) atomicAdd(gl + (size_t)(kl + tx) * C + c, acc);
                        acc = 0.f;
                        kl = m.l0;
                    }
                    if (on && (unsigned)(kl + tx) < (unsigned)Gl) {
                        const float w = tx ? m.lf : 1.f - m.lf;
                        acc = fmaf(dprod(s, coff + c) * Ps[s * ctot + coff + c], w, acc);
                    }
                }
                if (on && acc != 0.f) atomicAdd(gl + (size_t)(kl + tx) * C + c, acc);
            }
        }
        coff += C;
    }
}

// number of 64-lane iterations a pair needs
__device__ __forceinline__ int pair_iters(int model, const TfFactors& F) {
    int c = F.n_comp[0];
    if (model == TF_MODEL_VM) c = max(c, max(F.n_comp[1], F.n_comp[2]));
    return (4 * c + 63) >> 6;
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
