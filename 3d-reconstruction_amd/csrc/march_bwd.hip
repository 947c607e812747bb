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
//   pass 3  scatter  dL/dP = dL/df . (L m) m . w_tap,  dL/dL = dL/df . (P m) m . w_tap  with float atomics on
//           the channel-last gradient tensors.  Global float atomics on random 64..128-B pieces run far
//           below the streaming atomic rate (each piece is its own memory-side request), so the scatter is
//           run-length merged: consecutive samples of the ray that share a footprint are summed in
//           registers and leave as ONE atomic per lane (tf_device.h vm_chunk_*).  The small, heavily shared
//           line tensors are additionally spread over replicas (TfFactorGrads.n_rep).
#include "tf_device.h"

using namespace tf;

namespace {

struct BwdArgs {
    const float* grad_rgb_map;   // (R,3) dL/d rgb_map (after clamp)
    const float* rgb_pre;        // (R,3) pre-clamp rgb_map saved by the composite kernel
    const float* rgb;            // packed per-sample colours
    float* grad_rgb;             // packed dL/d colour (out)
    int white_bg;
    float* ent_xyz;              // binned mode: entry list (normalised xyz, dL/df) instead of scattering
    float* ent_df;
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
    const int app_keep = io.app_count[r];       // (a right-sized app list holds only the entries that fit: TfMarchIO.seg_cap)

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
            if (shade && appcnt + prefix_popc(ms) < app_keep) {
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
        // strictly later samples: the EXCLUSIVE scan value (the neighbour's inclusive one).  `incl - wdw` loses the
        // suffix to rounding when a nearly opaque sample's own term dwarfs everything behind it (saturated alpha:
        // a 0.7 % error in dL/dsigma, found by the random-configuration test with the ReLU activation).
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 0.f;
        const float suffix = carry + excl;
        carry += __shfl(incl, 63, 64);
        if (act) {
            const float dalpha = Tk * dw - suffix / t;
            sd[slot] = dalpha * df;
        }
    }
    __syncthreads();

    // ---------------- pass 3: scatter
    if (B.ent_xyz) {
        // binned mode: hand every valid sample (xyz, dL/df) to tf_binned_scatter through the sharded entry list
        const int shard = blockIdx.x & (TF_N_SHARDS - 1);
        const int worst = ((gridDim.x + TF_N_SHARDS - 1) / TF_N_SHARDS) * N;
        const int ent_cap = io.ent_seg_cap > 0 ? io.ent_seg_cap : worst;
        if (io.ent_offset) {      // the forward placed the entries (and wrote their coordinates) already
            const int base = io.ent_offset[r];
            const int keep = max(0, min(cnt, (shard + 1) * ent_cap - base));       // what fitted the shard (TfMarchIO.ent_seg_cap)
            for (int k = lane; k < keep; k += 64) B.ent_df[(size_t)base + k] = sd[k];
            return;
        }
        int base = 0, keep = 0;
        if (lane == 0) {
            const int a3 = atomicAdd(&io.counters[shard * TF_SHARD_STRIDE + 3], cnt);
            keep = max(0, min(cnt, ent_cap - a3));
            if (keep < cnt) atomicOr(&io.counters[TF_OVERFLOW_SLOT], 2);
            base = shard * ent_cap + min(a3, ent_cap);
        }
        base = __shfl(base, 0, 64);
        keep = __shfl(keep, 0, 64);
        for (int k = lane; k < keep; k += 64) {
            float p[3], u[3];
            sample_pos(ray, sample_z(F, ray, ztab, io.val_idx[vbase + k]), p);
            normalize(F, p, u);
            const size_t e = (size_t)base + k;
            B.ent_xyz[e * 3] = u[0];
            B.ent_xyz[e * 3 + 1] = u[1];
            B.ent_xyz[e * 3 + 2] = u[2];
            B.ent_df[e] = sd[k];
        }
        return;
    }
    if (F.model == TF_MODEL_VM) {
        // run-length merged scatter over chunks of 16 consecutive valid samples (tf_device.h)
        float* cbuf = smem + 3 * ncap;
        const int ctot = F.density.n_comp[0] + F.density.n_comp[1] + F.density.n_comp[2];
        for (int kb = 0; kb < cnt; kb += kChunk) {
            const int ns = min(kChunk, cnt - kb);
            auto u_of = [&](int s_, float* u) {
                float p[3];
                sample_pos(ray, sample_z(F, ray, ztab, io.val_idx[vbase + kb + s_]), p);
                normalize(F, p, u);
            };
            vm_chunk_gather(F.density, F.grid, ctot, ns, u_of, cbuf, lane);
            __syncthreads();
            auto dprod = [&](int s_, int) { return sd[kb + s_]; };
            vm_chunk_merge_scatter(F.density, G, F.grid, ctot, ns, dprod, cbuf, lane);
            __syncthreads();
        }
        return;
    }
    // CP: the whole wave on two samples at a time; the gather of the next pair is issued before the atomics
    // of the current one
    const int nit = pair_iters(F.model, F.density);
    PairVals cur, nxt;
    bool have = false;
    for (int kb = 0; kb < cnt; kb += 2) {
        const float dfA = sd[kb], dfB = kb + 1 < cnt ? sd[kb + 1] : 0.f;
        if (dfA == 0.f && dfB == 0.f) continue;
        float uA[3], uB[3], p[3];
        sample_pos(ray, sample_z(F, ray, ztab, io.val_idx[vbase + kb]), p);
        normalize(F, p, uA);
        sample_pos(ray, sample_z(F, ray, ztab, io.val_idx[vbase + (kb + 1 < cnt ? kb + 1 : kb)]), p);
        normalize(F, p, uB);
        auto dprod = [&](int s, int, int) { return s ? dfB : dfA; };
        for (int it = 0; it < nit; ++it) {
            cp_pair_gather(F.density, G, F.grid, uA, uB, dfA != 0.f, dfB != 0.f, dprod, lane, it, nxt);
            if (have) pair_commit(cur);
            cur = nxt;
            have = true;
        }
    }
    if (have) pair_commit(cur);
}

// One float4 per thread: element t of the packed side is quarter (t % q) of row idx[t / q] of the table.
template <bool SCATTER>
__global__ __launch_bounds__(256) void move_rows_kernel(float* __restrict__ table, const int* __restrict__ idx,
                                                        long long n4, int q, float* __restrict__ packed) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n4) return;
    const long long row = idx[t / q];
    float4* tp = reinterpret_cast<float4*>(table) + row * q + (t % q);
    float4* pp = reinterpret_cast<float4*>(packed) + t;
    if (SCATTER) *tp = *pp;
    else *pp = *tp;
}

__global__ __launch_bounds__(256) void reduce_replicas_kernel(const float* __restrict__ rep, int n_rep, int stride,
                                                              int numel, float* __restrict__ dst) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < numel; j += gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int r = 0; r < n_rep; ++r) a += rep[(size_t)r * stride + j];
        dst[j] = a;
    }
}

}  // namespace

extern "C" {

int tf_gather_rows(const float* table, const int* idx, int n_rows, int w, float* packed, tf_stream_t stream) {
    if (n_rows <= 0) return 0;
    if (w <= 0 || (w & 3)) return (int)hipErrorInvalidValue;
    const long long n4 = (long long)n_rows * (w / 4);
    hipLaunchKernelGGL((move_rows_kernel<false>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       const_cast<float*>(table), idx, n4, w / 4, packed);
    return TF_CHECK_LAUNCH();
}

int tf_scatter_rows(float* table, const int* idx, int n_rows, int w, const float* packed, tf_stream_t stream) {
    if (n_rows <= 0) return 0;
    if (w <= 0 || (w & 3)) return (int)hipErrorInvalidValue;
    const long long n4 = (long long)n_rows * (w / 4);
    hipLaunchKernelGGL((move_rows_kernel<true>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       table, idx, n4, w / 4, const_cast<float*>(packed));
    return TF_CHECK_LAUNCH();
}

int tf_reduce_replicas(const float* rep, int n_rep, int stride, int numel, float* dst, tf_stream_t stream) {
    if (numel <= 0) return 0;
    hipLaunchKernelGGL(reduce_replicas_kernel, dim3((numel + 255) / 256), dim3(256), 0, (hipStream_t)stream, rep, n_rep,
                       stride, numel, dst);
    return TF_CHECK_LAUNCH();
}

int tf_march_backward(const TfField* field, const TfMarchIO* io, const float* grad_rgb_map, const float* rgb_pre,
                      int white_bg, const float* rgb, float* grad_rgb, const TfFactorGrads* dgrads,
                      float* ent_xyz, float* ent_df, tf_stream_t stream) {
    if (io->n_rays <= 0) return 0;
    if (io->n_samples <= 0 || io->n_samples > TF_MAX_SAMPLES) return (int)hipErrorInvalidValue;
    const int ncap = (io->n_samples + 63) & ~63;
    const int ctot = field->density.n_comp[0] + field->density.n_comp[1] + field->density.n_comp[2];
    const size_t lds = (size_t)ncap * 12 + (field->model == TF_MODEL_VM ? (size_t)chunk_lds_words(ctot) * 4 : 0);
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
    BwdArgs B{grad_rgb_map, rgb_pre, rgb, grad_rgb, white_bg, ent_xyz, ent_df};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(march_backward_kernel), (size_t)(lds));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(march_backward_kernel, dim3(io->n_rays), dim3(64), lds, (hipStream_t)stream, *field, *io, B,
                       *dgrads);
    return TF_CHECK_LAUNCH();
}

#ifdef TF_PHASE_TIMING
int tf_debug_set_flags_march(int flags) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(tf_dbg_flags), &flags, sizeof(int));
}
#endif

}  // extern "C"
