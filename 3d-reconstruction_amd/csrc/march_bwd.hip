// march_bwd.hip — backward of compositing + density lookup (autograd of tensorBase.py:360-388
// and tensoRF.py:207-227 / :358-386).   gfx950, wave64.
//
// One wavefront per ray, working on the valid-sample list the forward saved (sample index + density
// feature):
//   pass 1  re-run the forward scan (same code, same order -> identical alpha / T / w and the same
//           shaded-sample slots), form dL/dw_k = g . (c_k - bg) and emit dL/dc_k = g w_k for the
//           shading backward;
//   pass 2  reverse suffix scan:  dL/dalpha_k = T_k dL/dw_k - (sum_{j>k} w_j dL/dw_j) / (1 - alpha_k + 1e-10),
//           then dL/dsigma_k = dL/dalpha_k . delta_k s (1 - alpha_k) and dL/df_k through the activation;
//   pass 3  4 lanes per sample re-gather the plane/line values and scatter-add
//           dL/dP = dL/df . (L m) m . w_tap,  dL/dL = dL/df . (P m) m . w_tap  with float atomics on the
//           channel-last gradient tensors (64-B contiguous per 4-lane group and tap).
#include "tf_device.h"

using namespace tf;

namespace {

__device__ __forceinline__ void atomic_add4(float* p, const float4_t& v) {
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(p + k, v[k]);
}

// scatter of one sample's density-feature gradient; lane `sub` of the 4-lane group covers channel quads
// sub, sub+4, ...
__device__ __forceinline__ void density_scatter(int model, const TfFactors& D, const TfFactorGrads& G,
                                                const int grid[3], const float u[3], int sub, float df) {
    if (model == TF_MODEL_VM) {
        VmTaps t;
        make_vm_taps(grid, u, t);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int C = D.n_comp[i];
            const float* mk = D.mask[i];
            float* gp = G.plane[i];
            float* gl = G.line[i];
            const Tap2& tp = t.p[i];
            const Tap1& tl = t.l[i];
            if ((C & 3) == 0) {
                for (int q = sub; q < (C >> 2); q += 4) {
                    const int ch = q * 4;
                    float4_t p = bilerp4(D.plane[i], C, tp, ch);
                    float4_t l = lerp4(D.line[i], C, tl, ch);
                    float4_t m2 = {1.f, 1.f, 1.f, 1.f};
                    if (mk) {
                        float4_t m = ld4(mk + ch);
                        m2 = m * m;                       // (P m)(L m): mask enters squared (tensoRF.py:225)
                    }
                    const float4_t gP = l * m2 * df, gL = p * m2 * df;
                    if (tp.w00 != 0.f) atomic_add4(gp + (size_t)tp.o00 * C + ch, gP * tp.w00);
                    if (tp.w01 != 0.f) atomic_add4(gp + (size_t)tp.o01 * C + ch, gP * tp.w01);
                    if (tp.w10 != 0.f) atomic_add4(gp + (size_t)tp.o10 * C + ch, gP * tp.w10);
                    if (tp.w11 != 0.f) atomic_add4(gp + (size_t)tp.o11 * C + ch, gP * tp.w11);
                    if (tl.w0 != 0.f) atomic_add4(gl + (size_t)tl.o0 * C + ch, gL * tl.w0);
                    if (tl.w1 != 0.f) atomic_add4(gl + (size_t)tl.o1 * C + ch, gL * tl.w1);
                }
            } else {
                for (int c = sub; c < C; c += 4) {
                    const float p = bilerp1(D.plane[i], C, tp, c), l = lerp1(D.line[i], C, tl, c);
                    const float m2 = mk ? mk[c] * mk[c] : 1.f;
                    const float gP = l * m2 * df, gL = p * m2 * df;
                    if (tp.w00 != 0.f) atomicAdd(gp + (size_t)tp.o00 * C + c, gP * tp.w00);
                    if (tp.w01 != 0.f) atomicAdd(gp + (size_t)tp.o01 * C + c, gP * tp.w01);
                    if (tp.w10 != 0.f) atomicAdd(gp + (size_t)tp.o10 * C + c, gP * tp.w10);
                    if (tp.w11 != 0.f) atomicAdd(gp + (size_t)tp.o11 * C + c, gP * tp.w11);
                    if (tl.w0 != 0.f) atomicAdd(gl + (size_t)tl.o0 * C + c, gL * tl.w0);
                    if (tl.w1 != 0.f) atomicAdd(gl + (size_t)tl.o1 * C + c, gL * tl.w1);
                }
            }
        }
    } else {
        // CP: f = sum_c L0 L1 L2 m  (tensoRF.py:363-384)
        const int C = D.n_comp[0];
        Tap1 t[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = make_tap1(u[vecm(i)], grid[vecm(i)]);
        const float* mk = D.mask[0];
        for (int c = sub; c < C; c += 4) {
            float l[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) l[i] = lerp1(D.line[i], C, t[i], c);
            const float m = mk ? mk[c] : 1.f;
            const float g0 = l[1] * l[2] * m * df, g1 = l[0] * l[2] * m * df, g2 = l[0] * l[1] * m * df;
            const float g[3] = {g0, g1, g2};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (t[i].w0 != 0.f) atomicAdd(G.line[i] + (size_t)t[i].o0 * C + c, g[i] * t[i].w0);
                if (t[i].w1 != 0.f) atomicAdd(G.line[i] + (size_t)t[i].o1 * C + c, g[i] * t[i].w1);
            }
        }
    }
}

struct BwdArgs {
    const float* grad_rgb_map;   // (R,3) dL/d rgb_map (after clamp)
    const float* rgb_pre;        // (R,3) pre-clamp rgb_map saved by the composite kernel
    const float* rgb;            // packed per-sample colours
    float* grad_rgb;             // packed dL/d colour (out)
    int white_bg;
};

__global__ __launch_bounds__(64) void march_backward_kernel(const TfField F, const TfMarchIO io, const BwdArgs B,
                                                            const TfFactorGrads G) {
    extern __shared__ float smem[];
    const int N = io.n_samples;
    const int ncap = (N + 63) & ~63;
    float* st = smem;              // t_k = 1 - alpha_k + 1e-10
    float* sT = smem + ncap;       // T_k
    float* sd = smem + 2 * ncap;   // dL/dw_k, later dL/df_k
    const int lane = threadIdx.x;
    const int r = xcd_remap(blockIdx.x, gridDim.x);
    const int cnt = io.val_count[r];
    if (cnt == 0) return;

    Ray ray;
    {
        const float* rp = io.rays + (size_t)r * 6;
#pragma unroll
        for (int a = 0; a < 3; ++a) { ray.o[a] = rp[a]; ray.d[a] = rp[3 + a]; }
        ray.jit = 0.f; ray.tmin = 0.f; ray.dnorm = 1.f;
        if (io.ndc) {
            float s = ray.d[0] * ray.d[0];
            s = s + ray.d[1] * ray.d[1];
            s = s + ray.d[2] * ray.d[2];
            ray.dnorm = sqrtf(s);
        } else {
            ray.tmin = slab_tmin(F, ray);
            if (io.jitter) ray.jit = io.jitter[r];
        }
    }
    const float* ztab = io.ndc ? io.z_table : nullptr;
    const size_t vbase = (size_t)r * N;

    // clamp passes the gradient inside [0,1] including the bounds (tensorBase.py:384)
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float pre = B.rgb_pre[(size_t)r * 3 + c];
        g[c] = (pre >= 0.f && pre <= 1.f) ? B.grad_rgb_map[(size_t)r * 3 + c] : 0.f;
    }
    const float bg = B.white_bg ? 1.f : 0.f;
    const int app_base = io.app_offset[r];

    // ---------------- pass 1: forward re-scan
    float T = 1.f;
    int appcnt = 0;
    for (int kb = 0; kb < cnt; kb += 64) {
        const int slot = kb + lane;
        const bool act = slot < cnt;
        float alpha = 0.f, t = 1.f;
        if (act) {
            const int idx = io.val_idx[vbase + slot];
            const float f = io.val_feat[vbase + slot];
            const float z = sample_z(F, ray, ztab, idx);
            const float sigma = density_act(F, f);
            float dist = 0.f;
            if (idx + 1 < N) dist = sample_z(F, ray, ztab, idx + 1) - z;
            if (io.ndc) dist = dist * ray.dnorm;
            const float ds = dist * F.distance_scale;
            alpha = 1.f - expf(-sigma * ds);
            t = (1.f - alpha) + 1e-10f;
        }
        const float incl = wave_scan_mul(t);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.f;
        const float Tk = T * excl;
        const float w = alpha * Tk;
        T = T * __shfl(incl, 63, 64);
        const bool shade = act && (w > F.weight_thres);
        const uint64_t ms = __ballot(shade);
        float dw = 0.f;
        if (act) {
            float c3[3] = {0.f, 0.f, 0.f};
            if (shade) {
                const size_t s = (size_t)app_base + appcnt + prefix_popc(ms);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    c3[c] = B.rgb[s * 3 + c];
                    B.grad_rgb[s * 3 + c] = g[c] * w;             // dL/dc_k = g w_k
                }
            }
            dw = g[0] * (c3[0] - bg) + g[1] * (c3[1] - bg) + g[2] * (c3[2] - bg);
            st[slot] = t;
            sT[slot] = Tk;
            sd[slot] = dw;
        }
        appcnt += __popcll(ms);
    }
    __syncthreads();

    // ---------------- pass 2: reverse suffix scan -> dL/df
    float carry = 0.f;   // sum_{j > current chunk} w_j dw_j
    const int last_kb = ((cnt - 1) >> 6) << 6;
    for (int kb = last_kb; kb >= 0; kb -= 64) {
        const int slot = kb + (63 - lane);    // lane 0 handles the LAST sample of the chunk
        const bool act = slot < cnt;
        float t = 1.f, Tk = 0.f, dw = 0.f, df = 0.f, wdw = 0.f;
        if (act) {
            t = st[slot]; Tk = sT[slot]; dw = sd[slot];
            const int idx = io.val_idx[vbase + slot];
            const float f = io.val_feat[vbase + slot];
            const float z = sample_z(F, ray, ztab, idx);
            const float sigma = density_act(F, f);
            float dist = 0.f;
            if (idx + 1 < N) dist = sample_z(F, ray, ztab, idx + 1) - z;
            if (io.ndc) dist = dist * ray.dnorm;
            const float ds = dist * F.distance_scale;
            const float e = expf(-sigma * ds);             // d alpha / d sigma = ds * exp(-sigma ds)
            wdw = ((1.f - e) * Tk) * dw;
            df = ds * e * density_act_grad(F, f);          // d alpha / d f
        }
        const float incl = wave_scan_add(wdw);     // lanes are in reverse sample order -> inclusive suffix
        const float suffix = carry + (incl - wdw); // strictly later samples
        carry += __shfl(incl, 63, 64);
        if (act) {
            const float dalpha = Tk * dw - suffix / t;
            sd[slot] = dalpha * df;
        }
    }
    __syncthreads();

    // ---------------- pass 3: scatter, 4 lanes per sample
    for (int kb = 0; kb < cnt; kb += 16) {
        const int slot = kb + (lane >> 2);
        if (slot < cnt) {
            const float df = sd[slot];
            if (df != 0.f) {
                const int idx = io.val_idx[vbase + slot];
                float p[3], u[3];
                sample_pos(ray, sample_z(F, ray, ztab, idx), p);
                normalize(F, p, u);
                density_scatter(F.model, F.density, G, F.grid, u, lane & 3, df);
            }
        }
    }
}

}  // namespace

extern "C" {

int tf_march_backward(const TfField* field, const TfMarchIO* io, const float* grad_rgb_map, const float* rgb_pre,
                      int white_bg, const float* rgb, float* grad_rgb, const TfFactorGrads* dgrads,
                      tf_stream_t stream) {
    if (io->n_rays <= 0) return 0;
    if (io->n_samples <= 0 || io->n_samples > TF_MAX_SAMPLES) return (int)hipErrorInvalidValue;
    const int ncap = (io->n_samples + 63) & ~63;
    const size_t lds = (size_t)ncap * 12;
    BwdArgs B{grad_rgb_map, rgb_pre, rgb, grad_rgb, white_bg};
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(march_backward_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(march_backward_kernel, dim3(io->n_rays), dim3(64), lds, (hipStream_t)stream, *field, *io, B,
                       *dgrads);
    return TF_CHECK_LAUNCH();
}

}  // extern "C"
