// march.hip — ray marching: sampling, bbox/alpha-mask tests, wave-level compaction, VM/CP density
// lookup, transmittance scan, weight threshold and the per-ray reductions.   gfx950, wave64.
//
// One wavefront (= one 64-thread workgroup) owns one ray:
//   phase A  every lane tests one sample per iteration (position, bbox, 1-byte alpha-cell lookup);
//            survivors are compacted in order into an LDS queue with ballot + mbcnt.
//   phase B  the queue is consumed 64 entries at a time; 4 lanes cooperate on one sample so that each
//            bilinear tap of the channel-last factor tensors is ONE contiguous 64-B segment.
//   phase C  one lane per queue entry: activation, alpha, wave product-scan of the transmittance,
//            weights, acc/depth partial sums, `w > thres` ballot -> shaded entries are staged in the
//            already-consumed part of the LDS queue.
//   end      one atomic per ray reserves a contiguous range of the packed app list (64 counter shards
//            on separate cache lines, so the reservation rate never limits the kernel).
// Replaces tensorBase.py:178-208 (sampling), :349-354 (alpha mask), :360-370 (density, raw2alpha,
// app_mask), :377,:386-388 (acc/depth) and tensoRF.py:207-227 / :358-386.
#include "tf_device.h"

using namespace tf;

namespace {

constexpr int kShards = TF_N_SHARDS;
constexpr int kShardStride = TF_SHARD_STRIDE;

__device__ __forceinline__ Ray load_ray(const TfField& F, const TfMarchIO& io, int r) {
    Ray ray;
    const float* rp = io.rays + (size_t)r * 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        ray.o[a] = rp[a];
        ray.d[a] = rp[3 + a];
    }
    ray.jit = 0.f;
    ray.tmin = 0.f;
    ray.dnorm = 1.f;
    if (io.ndc) {
        // torch.norm(viewdirs, dim=-1): sqrt(x*x + y*y + z*z)   (tensorBase.py:341)
        float s = ray.d[0] * ray.d[0];
        s = s + ray.d[1] * ray.d[1];
        s = s + ray.d[2] * ray.d[2];
        ray.dnorm = sqrtf(s);
    } else {
        ray.tmin = slab_tmin(F, ray);
        if (io.jitter) ray.jit = io.jitter[r];
    }
    return ray;
}

__global__ __launch_bounds__(64) void march_forward_kernel(const TfField F, const TfMarchIO io) {
    extern __shared__ float smem[];
    const int N = io.n_samples;
    const int ncap = (N + 63) & ~63;
    float* wq = smem;                                  // staged weights of shaded samples
    float* fbuf = smem + ncap;                         // density features of the current 64 entries
    uint16_t* q = reinterpret_cast<uint16_t*>(fbuf + 64);  // compacted sample indices (the density samples, in ray order)
    uint16_t* qs = q + ncap;                               // ... of the shaded ones

    const int lane = threadIdx.x;
    const int r = xcd_remap(blockIdx.x, gridDim.x);
    const int shard = blockIdx.x & (kShards - 1);
    const Ray ray = load_ray(F, io, r);
    const float* ztab = io.ndc ? io.z_table : nullptr;
    const int words = (N + 63) >> 6;

    // ---------------- phase A: validity + compaction
    // Samples behind the ray's exit from the box cannot be valid (tensorBase.py:206): the walk stops two steps past the
    // slab exit (two steps of slack >> the rounding of p = o + d z against the slab arithmetic; axes with d == 0 never
    // exit).  At config 2 that is 569 of 1039 slots.  NDC rays keep the full walk (their table spans near..far).
    int n_walk = N;
    if (!io.ndc) {
        float t_exit = INFINITY;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (ray.d[a] != 0.f) {
                const float ta = (F.aabb_hi[a] - ray.o[a]) / ray.d[a], tb = (F.aabb_lo[a] - ray.o[a]) / ray.d[a];
                t_exit = fminf(t_exit, fmaxf(ta, tb));
            }
        }
        const float span = (t_exit - ray.tmin) / F.step;            // NaN / inf -> full walk
        if (span < (float)N) n_walk = span > 0.f ? min(N, (int)span + 3) : min(N, 3);
    }
    if (io.dbg_z)          // tests: the sample positions along the ray, bit for bit (tensorBase.py:198-203)
        for (int i = lane; i < N; i += 64) io.dbg_z[(size_t)r * N + i] = sample_z(F, ray, ztab, i);
    if (lane == 0 && (io.dbg_bbox_bits || io.dbg_valid_bits))
        for (int wd = (n_walk + 63) >> 6; wd < words; ++wd) {        // words the shortened walk does not reach
            if (io.dbg_bbox_bits) io.dbg_bbox_bits[(size_t)r * words + wd] = 0;
            if (io.dbg_valid_bits) io.dbg_valid_bits[(size_t)r * words + wd] = 0;
        }
    int cnt = 0, nbbox = 0;
    TF_T0();
    // Four blocks of 64 slots per trip: their alpha-cell loads are requested together (one block per trip made the walk a
    // chain of ~9 dependent memory latencies: 64 % of a wave's time in this kernel at config 2).
    constexpr int U = 4;
    for (int base = 0; base < n_walk; base += 64 * U) {
        bool inb[U], chk[U];
        size_t cell[U];
        uint32_t allow[U], m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + 64 * u + lane;
            inb[u] = chk[u] = false;
            cell[u] = 0;
            allow[u] = 0;
            if (i < n_walk) {
                float z = sample_z(F, ray, ztab, i);
                float p[3];
                sample_pos(ray, z, p);
                inb[u] = in_bbox(F, p);
                if (inb[u] && F.alpha_cells) chk[u] = alpha_cell(F, p, cell[u], allow[u]);
            }
        }
        if (F.alpha_cells) {
#pragma unroll
            for (int u = 0; u < U; ++u) m[u] = F.alpha_cells[cell[u]];      // (cell 0 for the slots that need none)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + 64 * u >= n_walk) break;                            // wave-uniform
            const bool val = inb[u] && (F.alpha_cells == nullptr || (chk[u] && (m[u] & allow[u]) != 0));
            const uint64_t mb = __ballot(inb[u]), mv = __ballot(val);
            if (lane == 0) {
                if (io.dbg_bbox_bits) io.dbg_bbox_bits[(size_t)r * words + ((base >> 6) + u)] = mb;
                if (io.dbg_valid_bits) io.dbg_valid_bits[(size_t)r * words + ((base >> 6) + u)] = mv;
            }
            if (val) q[cnt + prefix_popc(mv)] = (uint16_t)(base + 64 * u + lane);
            cnt += __popcll(mv);
            nbbox += __popcll(mb);
        }
    }
    __syncthreads();
    TF_MARK(0);

    // ---------------- phases B + C over the compacted queue
    float T = 1.f, acc_l = 0.f, dep_l = 0.f;
    int appcnt = 0, done = 0;
    const size_t vbase = (size_t)r * N;
    for (int kb = 0; kb < cnt; kb += 64) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int slot = kb + s * 16 + (lane >> 2);
            float part = 0.f;
            if (slot < cnt) {
                const int idx = q[slot];
                float p[3], u[3];
                sample_pos(ray, sample_z(F, ray, ztab, idx), p);
                normalize(F, p, u);
                part = density_partial(F.model, F.density, F.grid, u, lane & 3);
            }
            const float f = quad_sum(part);
            if ((lane & 3) == 0) fbuf[s * 16 + (lane >> 2)] = f;
        }
        __syncthreads();
        TF_MARK(1);

        const int slot = kb + lane;
        const bool act = slot < cnt;
        const int idx = act ? (int)q[slot] : 0;
        const float f = fbuf[lane];
        const float z = sample_z(F, ray, ztab, idx);
        float alpha = 0.f, t = 1.f;
        if (act) {
            const float sigma = density_act(F, f);
            // dists = z[i+1]-z[i], last sample 0 (tensorBase.py:340,346); NDC scales by |d| (:342)
            float dist = 0.f;
            if (idx + 1 < N) dist = sample_z(F, ray, ztab, idx + 1) - z;
            if (io.ndc) dist = dist * ray.dnorm;
            const float ds = dist * F.distance_scale;
            alpha = 1.f - expf(-sigma * ds);                 // tensorBase.py:23
            t = (1.f - alpha) + 1e-10f;                       // tensorBase.py:25
        }
        const float incl = wave_scan_mul(t);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.f;
        const float w = alpha * (T * excl);                   // tensorBase.py:27
        T = T * __shfl(incl, 63, 64);
        acc_l += w;
        dep_l += w * z;
        const bool shade = act && (w > F.weight_thres);       // tensorBase.py:370
        const uint64_t ms = __ballot(shade);
        if (shade) {
            const int j = appcnt + prefix_popc(ms);
            qs[j] = (uint16_t)idx;
            wq[j] = w;
            if (io.dbg_app_bits)
                atomicOr(reinterpret_cast<unsigned*>(io.dbg_app_bits) + (size_t)r * words * 2 + (idx >> 5),
                         1u << (idx & 31));
        }
        appcnt += __popcll(ms);
        if (io.save_valid && act) {
            io.val_idx[vbase + slot] = idx;
            io.val_feat[vbase + slot] = f;
        }
        done = min(cnt, kb + 64);
        __syncthreads();
        TF_MARK(2);
        if (T < io.t_stop) break;                             // wave-uniform
    }

    // ---------------- per-ray results + packed app list
    const float acc = wave_sum(acc_l);
    float dep = wave_sum(dep_l);
    dep = dep + (1.f - acc) * io.rays[(size_t)r * 6 + 5];     // tensorBase.py:388 (last ray column, d_z)
    int base = 0, eb = 0, keep_a = 0, keep_e = 0;
    if (lane == 0) {
        int* ctr = io.counters + shard * kShardStride;
        const int worst = ((gridDim.x + kShards - 1) / kShards) * N;
        const int seg_cap = io.seg_cap > 0 ? io.seg_cap : worst, ent_cap = io.ent_seg_cap > 0 ? io.ent_seg_cap : worst;
        // the two reservations are requested together (one after the other they were two memory round trips in a row)
        const int a0 = appcnt ? atomicAdd(&ctr[0], appcnt) : 0;
        const int a3 = (io.ent_xyz && done) ? atomicAdd(&ctr[3], done) : 0;
        // right-sized lists: what does not fit the shard is dropped (the counters keep the demand) and the step is flagged
        keep_a = max(0, min(appcnt, seg_cap - a0));
        keep_e = io.ent_xyz ? max(0, min(done, ent_cap - a3)) : 0;
        const int over = (keep_a < appcnt ? 1 : 0) | ((io.ent_xyz && keep_e < done) ? 2 : 0);
        if (over) atomicOr(&io.counters[TF_OVERFLOW_SLOT], over);
        base = shard * seg_cap + min(a0, seg_cap);
        eb = shard * ent_cap + min(a3, ent_cap);
        const int a1 = atomicAdd(&ctr[1], done);
        // (without early sorting the backward reserves the density entries — one per valid sample: the same demand)
        if (io.save_valid && a1 + done > ent_cap) atomicOr(&io.counters[TF_OVERFLOW_SLOT], 2);
        atomicAdd(&ctr[2], nbbox);
        io.acc[r] = acc;
        io.depth[r] = dep;
        io.app_offset[r] = base;
        io.app_count[r] = keep_a;
        io.val_count[r] = done;
        if (io.ent_xyz) io.ent_offset[r] = eb;
    }
    if (io.ent_xyz) {      // density entry list of the binned backward scatter: coordinates now, dL/df in the backward
        eb = __shfl(eb, 0, 64);
        keep_e = __shfl(keep_e, 0, 64);
        for (int k = lane; k < keep_e; k += 64) {
            float p[3], u[3];
            sample_pos(ray, sample_z(F, ray, ztab, q[k]), p);      // (the queue still holds the density samples)
            normalize(F, p, u);
            const size_t e = (size_t)eb + k;
            io.ent_xyz[e * 3] = u[0];
            io.ent_xyz[e * 3 + 1] = u[1];
            io.ent_xyz[e * 3 + 2] = u[2];
        }
    }
    base = __shfl(base, 0, 64);
    keep_a = __shfl(keep_a, 0, 64);
    for (int j = lane; j < keep_a; j += 64) {
        const int idx = qs[j];
        float p[3], u[3];
        sample_pos(ray, sample_z(F, ray, ztab, idx), p);
        normalize(F, p, u);
        const size_t s = (size_t)base + j;
        io.app_ray[s] = r;
        io.app_xyz[s * 3 + 0] = u[0];
        io.app_xyz[s * 3 + 1] = u[1];
        io.app_xyz[s * 3 + 2] = u[2];
        io.app_w[s] = wq[j];
    }
    TF_MARK(3);
    TF_FLUSH();
}

// compute_densityfeature on an explicit list of normalised points (the public hook used by compute_alpha,
// tensorBase.py:311): 4 lanes per point, same gather code as the march kernel.
__global__ __launch_bounds__(256) void density_points_kernel(const TfField F, const float* __restrict__ xyz, int n,
                                                             float* __restrict__ out) {
    const int groups = (n + 63) / 64;
    for (int g = blockIdx.x * 4 + (threadIdx.x >> 6); g < groups; g += gridDim.x * 4) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pt = g * 64 + s * 16 + (lane >> 2);
            float part = 0.f;
            if (pt < n) {
                float u[3] = {xyz[(size_t)pt * 3], xyz[(size_t)pt * 3 + 1], xyz[(size_t)pt * 3 + 2]};
                part = density_partial(F.model, F.density, F.grid, u, lane & 3);
            }
            const float f = quad_sum(part);
            if (pt < n && (lane & 3) == 0) out[pt] = f;
        }
    }
}

// alpha volume (Gz,Gy,Gx) -> one byte per trilinear cell, cell index = floor coordinate + 1.
__global__ __launch_bounds__(256) void pack_alpha_cells_kernel(const float* __restrict__ vol, int gx, int gy, int gz,
                                                               uint8_t* __restrict__ cells) {
    const int sx = gx + 1, sy = gy + 1, sz = gz + 1;
    const size_t total = (size_t)sx * sy * sz;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (size_t)gridDim.x * blockDim.x) {
        const int cx = (int)(c % sx), cy = (int)((c / sx) % sy), cz = (int)(c / ((size_t)sx * sy));
        uint32_t m = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int x = cx - 1 + (b & 1), y = cy - 1 + ((b >> 1) & 1), z = cz - 1 + (b >> 2);
            if (x >= 0 && x < gx && y >= 0 && y < gy && z >= 0 && z < gz)
                if (vol[((size_t)z * gy + y) * gx + x] > 0.f) m |= 1u << b;
        }
        cells[c] = (uint8_t)m;
    }
}

__global__ __launch_bounds__(256) void pack_matrix_kernel(const float* __restrict__ src, int rows, int cols,
                                                          float* __restrict__ dst, int rows_pad, int kp) {
    const int total = rows_pad * kp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int rr = i / kp, c = i % kp;
        dst[i] = (rr < rows && c < cols) ? src[(size_t)rr * cols + c] : 0.f;
    }
}

__global__ __launch_bounds__(256) void pack_matrix_t_kernel(const float* __restrict__ src, int rows, int cols,
                                                            float* __restrict__ dst, int rows_pad, int kp) {
    const int total = kp * rows_pad;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i / rows_pad, r = i % rows_pad;
        dst[i] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
}

// several pack jobs in one launch (blockIdx.y = job): the training step refreshes 5 padded / transposed weight
// copies after every optimizer step
__global__ __launch_bounds__(256) void pack_matrices_kernel(const TfPackJob J) {
    pack_block(J, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

// loss = mean((a - b)^2) over n floats and grad = d loss / d a = 2 (a - b) / n, one workgroup (n is 3 x rays)
__global__ __launch_bounds__(1024) void mse_grad_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                        float grad_scale, float* __restrict__ loss,
                                                        float* __restrict__ grad) {
    __shared__ float red[16];
    const float inv = 1.f / (float)n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float d = a[i] - b[i];
        s = fmaf(d, d, s);
        grad[i] = 2.f * d * inv * grad_scale;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        *loss = t * inv;
    }
}

// rgb_map = sum_k w_k rgb_k (+ 1 - acc) clamped  (tensorBase.py:378-384).  8 lanes per ray; a ray's entries
// are contiguous and in sample order, so the summation order is fixed (deterministic).
__global__ __launch_bounds__(256) void composite_kernel(int n_rays, const int* __restrict__ app_offset,
                                                        const int* __restrict__ app_count,
                                                        const float* __restrict__ app_w,
                                                        const float* __restrict__ rgb, const float* __restrict__ acc,
                                                        int white_bg, float* __restrict__ rgb_map,
                                                        float* __restrict__ rgb_pre,
                                                        const int* __restrict__ counters,
                                                        long long* __restrict__ n_shaded, const TfLossFuse L,
                                                        const TfLive live) {
    if ((n_shaded || live.dev || live.host) && blockIdx.x == 0 && threadIdx.x < 64) {
        // num_valid_samples = app_mask.sum() (tensorBase.py:390); and the step's sample counts for the optimizer's gates
        // and the autograd binding: ray_valid.any() / app_mask.any() decide which parameters the reference's graph holds
        // (tensorBase.py:359, :370)
        long long shaded = counters[threadIdx.x * kShardStride], density = counters[threadIdx.x * kShardStride + 1];
        const int over = counters[TF_OVERFLOW_SLOT];
        static_assert(kShards == 64, "one shard per lane");
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            shaded += __shfl_xor(shaded, o, 64);
            density += __shfl_xor(density, o, 64);
        }
        if (threadIdx.x == 0) {
            if (n_shaded) *n_shaded = shaded;
            if (live.dev) {
                live.dev[0] = (float)density;
                live.dev[1] = (float)shaded;
                live.dev[2] = (float)over;
            }
            if (live.host) {
                const int sl = (live.slot && live.n_slots > 0) ? ((int)*live.slot % live.n_slots + live.n_slots) % live.n_slots : 0;
                int* h = live.host + 4 * sl;
                h[0] = (int)(density > 0x7fffffff ? 0x7fffffff : density);
                h[1] = (int)(shaded > 0x7fffffff ? 0x7fffffff : shaded);
                h[2] = over;
                __threadfence_system();
                h[3] = h[3] + 1;       // (number of reports this slot has taken: the host sees a step's words are in)
            }
        }
    }
    const int r = blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
    float c[3] = {0.f, 0.f, 0.f};
    // what the ray's result is combined with is requested up front, beside the sample loads (read where it is used, each
    // was one more memory round trip on a kernel that is little else)
    float acc_r = 0.f, tgt = 0.f;
    if (r < n_rays && sub < 3) {
        if (white_bg) acc_r = acc[r];
        if (L.target) tgt = L.target[(size_t)r * 3 + sub];
    }
    if (r < n_rays) {
        const int o = app_offset[r], n = app_count[r];
        for (int k = sub; k < n; k += 8) {
            const float w = app_w[o + k];
            const float* s = rgb + (size_t)(o + k) * 3;
            c[0] += w * s[0];
            c[1] += w * s[1];
            c[2] += w * s[2];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c[a] += __shfl_xor(c[a], 1, 64);
        c[a] += __shfl_xor(c[a], 2, 64);
        c[a] += __shfl_xor(c[a], 4, 64);
    }
    float d2 = 0.f;
    if (r < n_rays && sub < 3) {
        float v = sub == 0 ? c[0] : (sub == 1 ? c[1] : c[2]);
        if (white_bg) v += 1.f - acc_r;
        if (rgb_pre) rgb_pre[(size_t)r * 3 + sub] = v;
        const float o = fminf(fmaxf(v, 0.f), 1.f);
        rgb_map[(size_t)r * 3 + sub] = o;
        if (L.target) {      // loss = mean((rgb_map - target)^2) and its gradient (train.py:334), as mse_grad_kernel
            const float inv = 1.f / (float)(3 * n_rays);
            const float d = o - tgt;
            d2 = d * d;
            L.grad[(size_t)r * 3 + sub] = 2.f * d * inv * L.grad_scale;
        }
    }
    if (L.target) {
        // workgroup sum -> one atomic; the last workgroup to arrive publishes the mean and re-arms the two state words
        __shared__ float red[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d2 += __shfl_xor(d2, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d2;
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(L.state, (red[0] + red[1]) + (red[2] + red[3]));
            __threadfence();
            unsigned* arrivals = reinterpret_cast<unsigned*>(L.state + 1);
            if (atomicAdd(arrivals, 1u) == gridDim.x - 1) {
                __threadfence();
                const float total = atomicExch(L.state, 0.f);
                *L.loss = total * (1.f / (float)(3 * n_rays));
                *arrivals = 0u;
            }
        }
    }
}

}  // namespace

extern "C" {

#ifdef TF_PHASE_TIMING
int tf_debug_phase_cycles_march(unsigned long long* out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(tf_phase_cycles), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(tf_phase_cycles), z, sizeof(z));
    }
    return (int)e;
}
#endif


int tf_pack_alpha_cells(const float* volume, int gx, int gy, int gz, uint8_t* cells, tf_stream_t stream) {
    const size_t total = (size_t)(gx + 1) * (gy + 1) * (gz + 1);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_alpha_cells_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, volume, gx, gy, gz,
                       cells);
    return TF_CHECK_LAUNCH();
}

int tf_pack_matrix(const float* src, int rows, int cols, float* dst, int rows_pad, tf_stream_t stream) {
    const int kp = (cols + 15) & ~15;
    const int total = rows_pad * kp;
    hipLaunchKernelGGL(pack_matrix_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, rows,
                       cols, dst, rows_pad, kp);
    return TF_CHECK_LAUNCH();
}

int tf_pack_matrix_t(const float* src, int rows, int cols, float* dst, int rows_pad, tf_stream_t stream) {
    const int kp = (cols + 15) & ~15;
    const int total = kp * rows_pad;
    hipLaunchKernelGGL(pack_matrix_t_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, rows,
                       cols, dst, rows_pad, kp);
    return TF_CHECK_LAUNCH();
}

int tf_pack_matrices(const TfPackJob* job, tf_stream_t stream) {
    if (job->n < 1 || job->n > TF_PACK_MAX) return (int)hipErrorInvalidValue;
    const bool zero = job->n_zero > 0 && job->zero;
    hipLaunchKernelGGL(pack_matrices_kernel, dim3(pack_grid_x(*job), job->n + (zero ? 1 : 0)), dim3(256), 0,
                       (hipStream_t)stream, *job);
    return TF_CHECK_LAUNCH();
}

int tf_mse_grad(const float* a, const float* b, int n, float grad_scale, float* loss, float* grad, tf_stream_t stream) {
    if (n <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(mse_grad_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, n, grad_scale, loss, grad);
    return TF_CHECK_LAUNCH();
}

int tf_march_forward(const TfField* field, const TfMarchIO* io, tf_stream_t stream) {
    if (io->n_rays <= 0) return 0;
    if (io->n_samples <= 0 || io->n_samples > TF_MAX_SAMPLES) return (int)hipErrorInvalidValue;
    const int ncap = (io->n_samples + 63) & ~63;
    const size_t lds = (size_t)ncap * 8 + 256;      // weights (4 B), density queue and shaded queue (2 B each), 64 features
    const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(march_forward_kernel), lds);   // (> 64 KB at N = 8192)
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(march_forward_kernel, dim3(io->n_rays), dim3(64), lds, (hipStream_t)stream, *field, *io);
    return TF_CHECK_LAUNCH();
}

int tf_composite_forward(int n_rays, const int* app_offset, const int* app_count, const float* app_w,
                         const float* rgb, const float* acc, int white_bg, float* rgb_map, float* rgb_pre,
                         const int* counters, long long* n_shaded, const TfLive* live, tf_stream_t stream) {
    if (n_rays <= 0) return 0;
    hipLaunchKernelGGL(composite_kernel, dim3((n_rays + 31) / 32), dim3(256), 0, (hipStream_t)stream, n_rays,
                       app_offset, app_count, app_w, rgb, acc, white_bg, rgb_map, rgb_pre, counters, n_shaded,
                       TfLossFuse{nullptr, 0.f, nullptr, nullptr, nullptr}, live ? *live : TfLive{nullptr, nullptr, nullptr, 0, 0});
    return TF_CHECK_LAUNCH();
}

int tf_composite_forward_loss(int n_rays, const int* app_offset, const int* app_count, const float* app_w,
                              const float* rgb, const float* acc, int white_bg, float* rgb_map, float* rgb_pre,
                              const int* counters, long long* n_shaded, const TfLossFuse* fuse, const TfLive* live,
                              tf_stream_t stream) {
    if (n_rays <= 0) return 0;
    if (!fuse || !fuse->target || !fuse->grad || !fuse->loss || !fuse->state) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(composite_kernel, dim3((n_rays + 31) / 32), dim3(256), 0, (hipStream_t)stream, n_rays,
                       app_offset, app_count, app_w, rgb, acc, white_bg, rgb_map, rgb_pre, counters, n_shaded, *fuse,
                       live ? *live : TfLive{nullptr, nullptr, nullptr, 0, 0});
    return TF_CHECK_LAUNCH();
}

int tf_density_points(const TfField* field, const float* xyz_n, int n, float* out_f, tf_stream_t stream) {
    if (n <= 0) return 0;
    int blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(density_points_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *field, xyz_n, n,
                       out_f);
    return TF_CHECK_LAUNCH();
}

const char* tf_build_info(void) { return "tensorf_hip gfx950 wave64 fp32"; }

}  // extern "C"
