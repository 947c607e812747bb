// shade.hip — appearance lookup + shading head on the packed app list.   gfx950, wave64.
//
// Two kernels behind tf_shade_forward (launch_shade / launch_shade_pipe pick by the head's shape):
//   * shade_forward_pipe_kernel (further down): MLP heads with feature_c = 128 — one workgroup per CU, two crews of waves on
//     consecutive chunks, hidden layers on the bf16 matrix pipe with three-piece operands (fp32 accuracy);
//   * shade_forward_kernel (here): every other shape and the direct point-list entries.  A 512-thread workgroup (8 waves)
//     shades one tile of TF_TILE = 64 samples, two workgroups per CU:
//   1. gather: 8 lanes per sample read each bilinear tap of the channel-last appearance planes/lines as
//      contiguous 16-B pieces (128 B per 8-lane group per tap) and write the plane*line products
//      (sum n_comp wide) to an LDS tile V[64][.]                      tensoRF.py:238-260 / :394-410
//   2. basis_mat: feat^T = B . V^T on v_mfma_f32_16x16x4_f32 (exact fp32)  tensoRF.py:263
//   3. MLP input: [feat, view, PE blocks] built in LDS               mlp.py:8-13, 41-66
//   4. 2 hidden layers on the fp32 MFMA (weights streamed from L2, activations in LDS), ReLU fused
//      in the accumulator epilogue; output layer + sigmoid on the MFMA   mlp.py:34-38, 66-67
// Workgroups are persistent and walk the tiles of the 64 packed-list shards.
#include "tf_shade.h"

using namespace tf;

namespace {

// The packed sample list is cut EVENLY over the persistent workgroups (workgroup w owns samples [w q, (w+1) q) of the
// shards' concatenation, q = max(64, ceil(S / workgroups))) and walked in chunks of <= 64; a chunk may straddle a shard
// boundary, so it has two pieces.  (Round 1 handed out whole tiles through an atomic ticket: 11 k cycles per tile went
// into that dependent atomic + counter read, and nothing of the next tile could be requested ahead.)
struct FChunk {
    int s0, n0, s1, n1;
    __device__ __forceinline__ int n() const { return n0 + n1; }
    __device__ __forceinline__ size_t at(int k) const { return k < n0 ? (size_t)s0 + k : (size_t)s1 + (k - n0); }
};
__device__ __forceinline__ int fwd_samples_per_wg(int total, int n_wg) {
    const int q = (total + n_wg - 1) / n_wg;
    return q < M ? M : q;
}
// spre: exclusive prefix of the shards' sample counts (65 entries) — or NULL for a plain point list of n_direct entries
__device__ __forceinline__ bool fwd_locate(const TileSrc& src, const int* spre, int v, int v_end, FChunk& c) {
    c.s0 = c.n0 = c.s1 = c.n1 = 0;
    if (v >= v_end) return false;
    const int want = min(M, v_end - v);
    if (src.counters == nullptr) {
        c.s0 = v;
        c.n0 = want;
        return true;
    }
    static_assert(TF_N_SHARDS == 64, "one shard per lane");
    const int g = __builtin_popcountll(__ballot(spre[threadIdx.x & 63] <= v)) - 1;
    c.s0 = g * src.seg_cap + (v - spre[g]);
    c.n0 = min(want, spre[g + 1] - v);
    if (c.n0 < want) {
        int g2 = g + 1;
        while (g2 < TF_N_SHARDS && spre[g2 + 1] == spre[g2]) ++g2;
        if (g2 < TF_N_SHARDS) {
            c.s1 = g2 * src.seg_cap;
            c.n1 = min(want - c.n0, spre[g2 + 1] - spre[g2]);
        }
    }
    return true;
}

// training: copy the first n rows (w floats each, w a multiple of 4) of an LDS tile with row stride ld to the global rows
// at(row) * w, 16 B per lane
template <int NT, typename AtFn>
__device__ __forceinline__ void save_rows(float* dst, const float* tile, int ld, int w, int n, int tid, AtFn at) {
    const int w4 = w >> 2;
    const float inv = 1.f / (float)w4;
    for (int q = tid; q < n * w4; q += NT) {
        int row, c4;
        row_quad(q, w4, inv, row, c4);
        *reinterpret_cast<f32x4*>(dst + at(row) * (size_t)w + 4 * c4) = *reinterpret_cast<const f32x4*>(tile + row * ld + 4 * c4);
    }
}

// FT = feature_c / 16 hidden feature tiles (4, 8 or 16), NB = ceil(app_dim/16) basis feature tiles.
// 512 threads = 8 waves shade one 64-sample chunk; two workgroups share a CU (81,680 B of LDS each), i.e. 4 waves per
// SIMD (<= 128 VGPRs): the chain of dependent phases of one chunk is spread over 8 waves, and the two resident
// workgroups fill each other's barrier and memory waits (the gather is bound by the CU's fetch rate from the Infinity
// Cache, ~11-18 B/cycle/CU, whichever way its loads are issued).  The weight fragments of a hidden layer are requested
// one phase ahead (tf_shade.h load_a_frags), the sample info of the next chunk one chunk ahead.
// Wave w: feature tiles NFW*(w % FG) .. +NFW, sample tiles NSW*(w / FG) .. +NSW  (FG = 8 / SG feature groups).
template <int FT, int NB>
__global__ __launch_bounds__(512, 4) void shade_forward_kernel(const TfShade S, const TileSrc src, float* __restrict__ rgb_out,
                                                               float* __restrict__ feat_out, const TfShadeSave save) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // the only LDS object (16-B aligned base)
    constexpr int NT = 512, NW = 8, SG = FT < NW ? NW / FT : 1, FG = NW / SG, NFW = FT / FG, NSW = 4 / SG;
    constexpr bool PRE = NFW == 1;      // weight fragments of a whole layer fit the 128-register budget
    constexpr int KG1 = 12;             // layer 1: up to 12 k-groups (in_c <= 192)
    const ShadeLds L = shade_lds(S);
    int* pre = reinterpret_cast<int*>(lds + L.offPre);
    float* regA = lds + L.offA;   // V, then H1
    float* regB = lds + L.offB;   // X, then H2
    float* ixyz = lds + L.offInfo;           // [64][3]
    float* iview = lds + L.offInfo + 3 * M;  // [64][3]
    const int tid0 = threadIdx.x;

    int total = src.n_direct;
    if (src.counters) {   // sample prefix over shards (every workgroup computes the same table)
        shard_prefix(src.counters, src.seg_cap, pre, tid0);
        __syncthreads();
        total = pre[TF_N_SHARDS];
    }
    // The list is dealt evenly in units of 16 samples (one MFMA sample tile): workgroup b owns tiles [b T / W, (b + 1) T / W)
    // of the T = ceil(total / 16) and walks them in chunks of <= 64; its last chunk is usually a partial one, and a partial
    // chunk costs its active sample tiles only (`nt` below).  (Dealt in whole 64-sample chunks, 1268 chunks on 512
    // workgroups made three rounds of which the third was half empty: 2.48 chunk times of work took 3.  Handing a tail of
    // the list — any share, up to all of it — out by a ticket taken one chunk ahead changed nothing beside the training
    // step's sorts as they are, 0.714-0.725 ms per step against 0.714; and beside sorts with 1024-thread workgroups, which
    // finish in half the time but starve the shading workgroup on their CU (this kernel: 200 us instead of 126), it
    // recovered the loss only in part: 0.713 ms with everything by ticket, against 0.752 dealt evenly.)
    const long long n_tiles = (total + 15) / 16;
    const int v_begin = (int)(((long long)blockIdx.x * n_tiles) / (long long)gridDim.x) * 16;
    const int v_end = min(total, (int)((((long long)blockIdx.x + 1) * n_tiles) / (long long)gridDim.x) * 16);

    // per-sample info of thread tid < 64, fetched for the NEXT chunk while the current one is processed
    // (app_ray -> rays is a chain of two global latencies)
    float nx_x[3] = {0.f, 0.f, 0.f}, nx_v[3] = {0.f, 0.f, 0.f};
    auto fetch_info = [&](const FChunk& ck, int tid) {
#pragma unroll
        for (int a = 0; a < 3; ++a) nx_x[a] = nx_v[a] = 0.f;
        if (tid < ck.n()) {
            const size_t s = ck.at(tid);
            nx_x[0] = src.app_xyz[s * 3]; nx_x[1] = src.app_xyz[s * 3 + 1]; nx_x[2] = src.app_xyz[s * 3 + 2];
            if (src.rays) {
                const float* rp = src.rays + (size_t)src.app_ray[s] * 6 + 3;
                nx_v[0] = rp[0]; nx_v[1] = rp[1]; nx_v[2] = rp[2];
            } else if (src.view_direct) {
                nx_v[0] = src.view_direct[s * 3]; nx_v[1] = src.view_direct[s * 3 + 1]; nx_v[2] = src.view_direct[s * 3 + 2];
            }
        }
    };
    {
        FChunk c1;
        if (fwd_locate(src, pre, v_begin, v_end, c1)) fetch_info(c1, tid0);
    }

    TF_T0();
    for (int v = v_begin;;) {
        FChunk ck;
        if (!fwd_locate(src, pre, v, v_end, ck)) break;
        const int n = ck.n();
        v += n;
        const int nt = (n + 15) >> 4, n16 = 16 * nt;      // active sample tiles / rows of this chunk
        // thread coordinates from an opaque copy of the thread id, so that no per-thread address of a later phase is
        // computed (and kept in registers) outside the chunk loop: the kernel has 128 VGPRs at 4 waves per SIMD
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        auto at = [&](int r) { return ck.at(r); };
        TF_MARK(7);

        // ---- chunk info (requested during the previous chunk)
        if (tid < M) {
            if (src.ndc && tid < n) {   // viewdirs / rays_norm  (tensorBase.py:341-343)
                float q = nx_v[0] * nx_v[0];
                q = q + nx_v[1] * nx_v[1];
                q = q + nx_v[2] * nx_v[2];
                const float nrm = sqrtf(q);
                nx_v[0] = nx_v[0] / nrm; nx_v[1] = nx_v[1] / nrm; nx_v[2] = nx_v[2] / nrm;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                ixyz[tid * 3 + a] = nx_x[a];
                iview[tid * 3 + a] = nx_v[a];
            }
        }
        __syncthreads();
        {
            FChunk c1;
            if (fwd_locate(src, pre, v, v_end, c1)) fetch_info(c1, tid);
        }
        TF_MARK(0);

        // ---- 1. appearance gather -> V: 8 lanes per sample
        if (!src.feat_in) {
            const int smp = (tid >> 6) * 8 + (lane >> 3), sub = lane & 7;
            if (smp < n16) {
                float u[3] = {ixyz[smp * 3], ixyz[smp * 3 + 1], ixyz[smp * 3 + 2]};
                float* vrow = regA + smp * L.sv;
                if (!app_products_lanes8(S, u, sub, vrow)) app_products(S, u, sub, vrow, 8);
                for (int c = S.n_app_total + sub; c < kpad16(S.n_app_total); c += 8) vrow[c] = 0.f;
            }
        }
        __syncthreads();
        TF_MARK(1);
        if (save.v) {     // training: the product rows go back to HBM for the backward (dB = dfeat^T V), 16 B per lane
            const int nat = S.n_app_total;
            if ((nat & 3) == 0) {
                save_rows<NT>(save.v, regA, L.sv, nat, n, tid, at);
            } else {
                for (int smp = wave; smp < n; smp += NW)
                    for (int c = lane; c < nat; c += 64) save.v[at(smp) * nat + c] = regA[smp * L.sv + c];
            }
        }

        // ---- 2. basis: feat[s][f] = sum_k B[f][k] V[s][k]; (feature tile, sample tile) pairs dealt to the 8 waves
        if (src.feat_in) {      // renderModule(pts, viewdirs, features) alone: the caller's features are the MLP input
            for (int it = tid; it < M * S.app_dim; it += NT) {
                const int smp = it / S.app_dim, d = it - smp * S.app_dim;
                regB[smp * L.sx + d] = smp < n ? src.feat_in[at(smp) * S.app_dim + d] : 0.f;
            }
        }
        for (int pr = src.feat_in ? 4 * NB : wave; pr < 4 * NB; pr += NW) {
            const int bf = pr >> 2, bs = pr & 3;
            if (bs >= nt) continue;
            f32x4 acc[1][1];
            acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mma_block<1, 1>(S.basis, kpad16(S.n_app_total), 16 * bf, regA, L.sv, 16 * bs, kpad16(S.n_app_total) / 16, acc, lane);
            const int smp = 16 * bs + (lane & 15), g = lane >> 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = 16 * bf + 4 * g + e;
                if (f < S.app_dim) {
                    regB[smp * L.sx + f] = acc[0][0][e];
                    if (feat_out && smp < n) feat_out[at(smp) * S.app_dim + f] = acc[0][0][e];
                }
            }
        }
        if (feat_out) {      // compute_appfeature hook: features only
            __syncthreads();
            continue;
        }
        if (tid < M) {
#pragma unroll
            for (int a = 0; a < 3; ++a) regB[tid * L.sx + S.app_dim + a] = iview[tid * 3 + a];
        }
        __syncthreads();
        TF_MARK(2);

        if (S.head != TF_HEAD_MLP) {   // SHRender / RGBRender (mlp.py:15-25)
            if (tid < n * 3) {
                const int smp = tid / 3, ch = tid % 3;
                const float* x = regB + smp * L.sx;
                float o;
                if (S.head == TF_HEAD_RGB) {
                    o = x[ch];
                } else {
                    float y[9], d[3] = {iview[smp * 3], iview[smp * 3 + 1], iview[smp * 3 + 2]};
                    sh9(d, y);
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < 9; ++k) a += y[k] * x[ch * 9 + k];
                    o = fmaxf(a + 0.5f, 0.f);
                }
                rgb_out[at(smp) * 3 + ch] = o;
            }
            __syncthreads();
            continue;
        }

        // ---- 3. positional-encoding blocks + zero K padding
        {
            int off = S.app_dim + 3;
            for (int b = 0; b < S.n_pe; ++b) {
                const int src_k = S.pe[b].src, F = S.pe[b].freqs;
                const int D = src_k == TF_SRC_FEAT ? S.app_dim : 3;
                const float* mk = S.pe[b].mask;
                float* xb = regB;
                const int sx = L.sx;
                pe_block<NT>(regB, L.sx, off, D, F, mk, tid, [&](int smp, int d) {
                    return src_k == TF_SRC_FEAT ? xb[smp * sx + d] : (src_k == TF_SRC_VIEW ? iview[smp * 3 + d] : ixyz[smp * 3 + d]);
                }, n16);
                off += 2 * D * F;
            }
            const int kp = kpad16(S.in_c);
            for (int it = tid; it < n16 * 16; it += NT) {     // the K padding is < 16 columns
                const int smp = it >> 4, c = S.in_c + (it & 15);
                if (c < kp) regB[smp * L.sx + c] = 0.f;
            }
        }
        const int FC = S.feature_c, kp1 = kpad16(S.in_c), kt1 = kp1 / 16;
        const int f_base = 16 * NFW * (wave % FG), s_base = 16 * NSW * (wave / FG);
        const int lc = lane & 15, lg = lane >> 4;
        const int nst = min(NSW, max(0, nt - NSW * (wave / FG)));      // this wave's active sample tiles
        f32x4 fr1[PRE ? KG1 : 1][1];          // layer-1 weight fragments, in flight across the barrier
        if constexpr (PRE) load_a_frags<1, KG1>(S.w1, kp1, f_base, kt1, lane, fr1);
        __syncthreads();
        TF_MARK(3);
        if (save.x) save_rows<NT>(save.x, regB, L.sx, kp1, n, tid, at);   // training: the (zero padded) MLP input rows

        // ---- 4. hidden layers
        f32x4 fr2[PRE ? (FT <= 8 ? FT : 1) : 1][1];
        {
            f32x4 acc[NFW][NSW];
#pragma unroll
            for (int i = 0; i < NFW; ++i)
#pragma unroll
                for (int j = 0; j < NSW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (PRE) {
                // (a partial chunk: the wave multiplies the first half of its sample tiles only)
                if (NSW == 1 || 2 * nst > NSW) mma_frags<1, NSW, KG1>(fr1, regB, L.sx, s_base, kt1, acc, lane);
                else if (nst > 0) mma_frags<1, (NSW > 1 ? NSW / 2 : 1), KG1>(fr1, regB, L.sx, s_base, kt1, reinterpret_cast<f32x4 (&)[1][NSW > 1 ? NSW / 2 : 1]>(acc), lane);
                load_a_frags<1, FT>(S.w2, kpad16(FC), f_base, FT, lane, fr2);     // layer 2's weights: behind this epilogue
            } else {
                mma_block<NFW, NSW>(S.w1, kp1, f_base, regB, L.sx, s_base, kt1, acc, lane);
            }
#pragma unroll
            for (int i = 0; i < NFW; ++i) {
                const int f = f_base + 16 * i + 4 * lg;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(S.b1 + f);
#pragma unroll
                for (int j = 0; j < NSW; ++j) {
                    if (PRE && j >= nst) continue;
                    f32x4 h = acc[i][j] + bias;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = fmaxf(h[e], 0.f);
                    *reinterpret_cast<f32x4*>(regA + (s_base + 16 * j + lc) * L.sh + f) = h;
                }
            }
        }
        __syncthreads();
        TF_MARK(4);
        if (save.h1) save_rows<NT>(save.h1, regA, L.sh, FC, n, tid, at);
        // output-layer weights as MFMA fragments (rows 0..2 of a 16-row operand tile, the other rows zero) for the waves
        // that run the output layer (0..3)
        constexpr int KG3 = FT <= 8 ? FT : 8;
        f32x4 fr3[KG3];
        {
            f32x4 acc[NFW][NSW];
#pragma unroll
            for (int i = 0; i < NFW; ++i)
#pragma unroll
                for (int j = 0; j < NSW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (PRE) {
                if (NSW == 1 || 2 * nst > NSW) mma_frags<1, NSW, FT>(fr2, regA, L.sh, s_base, FT, acc, lane);
                else if (nst > 0) mma_frags<1, (NSW > 1 ? NSW / 2 : 1), FT>(fr2, regA, L.sh, s_base, FT, reinterpret_cast<f32x4 (&)[1][NSW > 1 ? NSW / 2 : 1]>(acc), lane);
            }
            else mma_block<NFW, NSW>(S.w2, kpad16(FC), f_base, regA, L.sh, s_base, FC / 16, acc, lane);
#pragma unroll
            for (int i = 0; i < NFW; ++i) {
                const int f = f_base + 16 * i + 4 * lg;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(S.b2 + f);
#pragma unroll
                for (int j = 0; j < NSW; ++j) {
                    if (PRE && j >= nst) continue;
                    f32x4 h = acc[i][j] + bias;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = fmaxf(h[e], 0.f);
                    *reinterpret_cast<f32x4*>(regB + (s_base + 16 * j + lc) * L.sh + f) = h;
                }
            }
        }
        __syncthreads();
        TF_MARK(5);
        if (save.h2) save_rows<NT>(save.h2, regB, L.sh, FC, n, tid, at);
        if (wave < nt) {     // all at once (unconditional loads, rows >= 3 zeroed after)
            const int r3 = lc < 3 ? lc : 2;
#pragma unroll
            for (int k = 0; k < KG3; ++k) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(S.w3 + r3 * FC + 16 * k + 4 * lg);
                fr3[k] = lc < 3 ? w : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- 5. output layer + sigmoid on the MFMA: o[c][s] = sum_f W3[c][f] H2[s][f], W3 as rows 0..2 of a 16-row
        // operand tile; wave w < 4 takes sample tile w (the other waves go on to the next chunk's info phase — nothing
        // they write there is read here)
        if (wave < nt) {
            const int r = lane & 15, kq = lane >> 4;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};      // two chains: no dependent-MFMA stalls
            const float* xp = regB + (16 * wave + r) * L.sh + 4 * kq;
#pragma unroll
            for (int k = 0; k < KG3; ++k) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(xp + 16 * k);
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fr3[k][e], b[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fr3[k][e + 1], b[e + 1], acc1, 0, 0, 0);
                }
            }
            for (int kg = KG3; kg < FC / 16; ++kg) {        // feature_c 256: the second half of the k-groups, streamed
                const int r3 = r < 3 ? r : 2;
                const f32x4 w = *reinterpret_cast<const f32x4*>(S.w3 + r3 * FC + 16 * kg + 4 * kq);
                const f32x4 a = r < 3 ? w : (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 b = *reinterpret_cast<const f32x4*>(xp + 16 * kg);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc0, 0, 0, 0);
            }
            const f32x4 acc = acc0 + acc1;
            // D[c = 4 lg + reg][s = lc]: lanes 0..15 hold the three channels of sample 16 wave + lc
            const int smp = 16 * wave + lc;
            if (lg == 0 && smp < n) {
                float* o = rgb_out + at(smp) * 3;
                o[0] = 1.f / (1.f + expf(-(acc[0] + S.b3[0])));
                o[1] = 1.f / (1.f + expf(-(acc[1] + S.b3[1])));
                o[2] = 1.f / (1.f + expf(-(acc[2] + S.b3[2])));
            }
        }
        TF_MARK(6);
        // no barrier: the next chunk's info phase writes ixyz / iview only and ends in a barrier before anything
        // touches the H2 region again
    }
    TF_FLUSH();
}

// ---- the pipelined forward (packed list, MLP heads with feature_c = 128): round 3 ------------------------------------
// Why: the kernel above spends a chunk in ~13 DEPENDENT global round trips (three tap stages, the basis and layer
// weights, biases, W3, b3), and on a CU whose memory pipe is busy with the other workgroup's gather (~220 KB per chunk at
// 11-18 B/cycle) every one of them queues behind that traffic: 65 k cycles per chunk and CU for 22 k cycles of MFMA
// issue.  Here ONE 768-thread workgroup owns the CU and its 12 waves (three per SIMD: 168 VGPRs) are two crews working on
// consecutive chunks, in lock step through five LDS-only barriers per iteration (s_barrier counts waves, whichever
// instruction they arrive at):
//   waves 0..7  (MLP crew),   chunk c:   P1 basis, view -> X   P2 encodings -> X   P3 X -> H1   P4 H1 -> H2   P5 output layer
//   waves 8..11 (gather crew), chunk c + 1: the nine (plane, channel-quad) units of the appearance gather -> V, dealt
//                             0 / 3 / 3 / 3 / 0 over the phases; coordinates and view directions of chunk c + 2
// The MLP crew keeps its slice of W1 and W2 (one 16-feature tile per wave) in registers for the whole launch and never
// waits for a load it has just issued; the gather — the only bulk fetch — runs beside everything else.
//
// The hidden layers and the output layer run on the bf16 matrix pipe at FULL fp32 accuracy: every operand is three
// bf16 pieces, x = h + m + l, each the next 8 bits of the significand (truncation: exact), and a product is the six terms
// hh + hm + mh + mm + hl + lh on v_mfma_f32_16x16x32_bf16 — 6 instructions of 4 passes per 32 k where the fp32 path
// issues 8 of 8 passes (2.7 x); the dropped terms ml, lm, ll are <= 2^-24 of |a||b| each: max error 1.4e-7 of the largest
// entry against 3.0e-7 for v_mfma_f32_16x16x4_f32 itself (profiles/r03_mfma_bf16x3_probe.txt).  The activations live in
// LDS as three bf16 planes (6 B per element — the exact value, h + m + l reproduces the fp32 the training rows need), the
// weights as three packed fragments per k-step in registers (108 VGPRs).  (On this hardware a wave that streams MFMAs
// blocks the VALU of the other waves on its SIMD — same probe — so the matrix time is not hidden behind anything: it had to
// shrink.)  The basis product (K = 144, 7 % of the flops) stays on the fp32 instruction.
//
// LDS: X planes (H2 overlays them), H1 planes, V, the sample info double-buffered, W3 planes, biases — 158 KB.  The k
// extents are fixed (160 / 144 / 128: in_c <= 160, n_app_total <= 144): shorter operands are zero padded, the MFMA loops
// carry no guards.  Shapes beyond that (or other heads / hidden widths) use the kernel above.
constexpr int PIPE_KX = 160, PIPE_KB = 144, PIPE_FC = 128;
constexpr int PIPE_XROW = 2 * PIPE_KX + 16, PIPE_HROW = 2 * PIPE_FC + 16;      // plane row strides in BYTES (conflict-free b128 reads)
constexpr int PIPE_XPLANE = M * PIPE_XROW, PIPE_HPLANE = M * PIPE_HROW;
struct PipeLds {      // byte offsets
    int sv;           // V row stride in floats
    int offX, offH1, offV, offInfo0, offInfo1, offW3, offBias, offDesc, offPre, total;
};
__host__ __device__ inline PipeLds pipe_lds() {
    PipeLds L;
    L.sv = PIPE_KB + 4;
    L.offX = 0;
    L.offH1 = 3 * PIPE_XPLANE;
    L.offV = L.offH1 + 3 * PIPE_HPLANE;
    L.offInfo0 = L.offV + M * L.sv * 4;
    L.offInfo1 = L.offInfo0 + M * 6 * 4;
    L.offW3 = L.offInfo1 + M * 6 * 4;                 // three planes of [3][128] bf16
    L.offBias = L.offW3 + 3 * 3 * PIPE_FC * 2;       // b1, b2, b3
    L.offDesc = L.offBias + (2 * PIPE_FC + 4) * 4;   // ring of 4 chunk descriptors (FChunk), written by the gather crew
    L.offPre = L.offDesc + 64;
    L.total = L.offPre + 68 * 4;
    return L;
}

#ifdef TF_PHASE_TIMING      // diagnostic build: 4 = no gather, 8 = no hidden-layer MFMAs, 16 = no basis / encodings, 32 = no output layer (timing only)
#define TF_ABL_INIT() const int abl_ = __builtin_amdgcn_readfirstlane(tf_dbg_flags)
#define TF_ABL(bit) (abl_ & (bit))
#define TF_PIPE_FLUSH(first_thread, arr) do { if ((int)threadIdx.x == (first_thread)) for (int _i = 0; _i < 16; ++_i) atomicAdd(&arr[_i], _ph[_i]); } while (0)
#else
#define TF_ABL_INIT()
#define TF_ABL(bit) 0
#define TF_PIPE_FLUSH(first_thread, arr)
#endif

// Units [U0, U0 + NU) of the appearance gather of one sample, by lane `sub` of its four: unit u = plane u / 3, channel quad
// sub + 4 (u % 3) (components a multiple of 4 and <= 48 per plane: pipe_gather_ok).  All 6 NU tap pieces are requested
// before the first is used; the products (P m)(L m) go to vrow (tensoRF.py:238-260).
template <int U0, int NU>
__device__ __forceinline__ void gather_units(const TfShade& S, const VmTaps& t, int sub, float* vrow) {
    float4_t pa[NU][4], la[NU][2];
    bool has[NU];
    int ch[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int i = (U0 + k) / 3, j = (U0 + k) % 3;
        const int C = S.app.n_comp[i], Q = C >> 2;
        has[k] = sub + 4 * j < Q;
        ch[k] = 4 * (has[k] ? sub + 4 * j : (Q > sub ? sub : 0));      // (lanes without a quad re-read one: no branch)
        pa[k][0] = ld4(S.app.plane[i] + (size_t)t.p[i].o00 * C + ch[k]);
        pa[k][1] = ld4(S.app.plane[i] + (size_t)t.p[i].o01 * C + ch[k]);
        pa[k][2] = ld4(S.app.plane[i] + (size_t)t.p[i].o10 * C + ch[k]);
        pa[k][3] = ld4(S.app.plane[i] + (size_t)t.p[i].o11 * C + ch[k]);
        la[k][0] = ld4(S.app.line[i] + (size_t)t.l[i].o0 * C + ch[k]);
        la[k][1] = ld4(S.app.line[i] + (size_t)t.l[i].o1 * C + ch[k]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int i = (U0 + k) / 3;
        if (!has[k]) continue;
        const int coff = i == 0 ? 0 : (i == 1 ? S.app.n_comp[0] : S.app.n_comp[0] + S.app.n_comp[1]);
        float4_t p, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            p[e] = fmaf(pa[k][3][e], t.p[i].w11, fmaf(pa[k][2][e], t.p[i].w10, fmaf(pa[k][1][e], t.p[i].w01, pa[k][0][e] * t.p[i].w00)));
            l[e] = fmaf(la[k][1][e], t.l[i].w1, la[k][0][e] * t.l[i].w0);
        }
        const float* mk = S.app.mask[i];
        if (mk) {
            const float4_t m = ld4(mk + ch[k]);
            p *= m;
            l *= m;
        }
        *reinterpret_cast<float4_t*>(vrow + coff + ch[k]) = p * l;
    }
}
__host__ __device__ inline bool pipe_gather_ok(const TfShade& S) {
    if (S.model != TF_MODEL_VM) return false;
    for (int i = 0; i < 3; ++i)
        if ((S.app.n_comp[i] & 3) != 0 || S.app.n_comp[i] > 48) return false;
    return true;
}
typedef int int4_desc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ FChunk desc_chunk(const int4_desc d) {      // wave-uniform copy (the fields index LDS / global rows)
    FChunk c;
    c.s0 = __builtin_amdgcn_readfirstlane(d[0]);
    c.n0 = __builtin_amdgcn_readfirstlane(d[1]);
    c.s1 = __builtin_amdgcn_readfirstlane(d[2]);
    c.n1 = __builtin_amdgcn_readfirstlane(d[3]);
    return c;
}
// fp32 weight fragments of a wave's feature tile (the basis product); k-groups past the matrix are ZERO
template <int KG>
__device__ __forceinline__ void load_resident_frags(const float* __restrict__ Wg, int ldw, int f_base, int kgroups, int lane,
                                                    f32x4 (&a)[KG][1]) {
    load_a_frags<1, KG>(Wg, ldw, f_base, kgroups > 0 ? kgroups : 1, lane, a);
#pragma unroll
    for (int kg = 0; kg < KG; ++kg)
        if (kg >= kgroups) a[kg][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// ---- three-piece bf16 operands -----------------------------------------------------------------------------------------
// x = h + m + l exactly; every piece is returned in the UPPER half of a word (its lower half is zero)
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(h);                  // exact
    m = __float_as_uint(r1) & 0xFFFF0000u;
    l = __float_as_uint(r1 - __uint_as_float(m));             // exact, <= 8 significant bits
}
struct Frag3 {      // eight consecutive k of one row, as the three A (or B) operands of the K = 32 instruction
    bf16x8 h, m, l;
};
__device__ __forceinline__ Frag3 split_frag(const float (&x)[8]) {
    u32x4 ph, pm, pl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h0, m0, l0, h1, m1, l1;
        split3(x[2 * i], h0, m0, l0);
        split3(x[2 * i + 1], h1, m1, l1);
        ph[i] = (h0 >> 16) | h1;
        pm[i] = (m0 >> 16) | m1;
        pl[i] = (l0 >> 16) | l1;
    }
    Frag3 f;
    f.h = __builtin_bit_cast(bf16x8, ph);
    f.m = __builtin_bit_cast(bf16x8, pm);
    f.l = __builtin_bit_cast(bf16x8, pl);
    return f;
}
// weight fragments of feature row f_base + (lane & 15) for KS k-steps of 32: W is row-major fp32 [.][ldw], zero beyond `cols`
template <int KS>
__device__ __forceinline__ void load_weight_frags3(const float* __restrict__ W, int ldw, int cols, int f_base, int lane, Frag3 (&a)[KS]) {
    const int r = lane & 15, kq = lane >> 4;
    const float* wp = W + (size_t)(f_base + r) * ldw + 8 * kq;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float x[8];
        const bool in = 32 * ks + 8 * kq < cols;       // (packed rows are padded to multiples of 16 with zeros: 8 k at a time are in or out)
        const f32x4 lo = in ? *reinterpret_cast<const f32x4*>(wp + 32 * ks) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 hi = in ? *reinterpret_cast<const f32x4*>(wp + 32 * ks + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[e] = lo[e];
            x[4 + e] = hi[e];
        }
        a[ks] = split_frag(x);
    }
}
// one element into the three planes of a tile (plane stride `plane`, row stride `row`, both in bytes)
__device__ __forceinline__ void put3(char* base, int plane, int row, int r, int c, float v) {
    unsigned h, m, l;
    split3(v, h, m, l);
    char* p = base + r * row + 2 * c;
    *reinterpret_cast<unsigned short*>(p) = (unsigned short)(h >> 16);
    *reinterpret_cast<unsigned short*>(p + plane) = (unsigned short)(m >> 16);
    *reinterpret_cast<unsigned short*>(p + 2 * plane) = (unsigned short)(l >> 16);
}
__device__ __forceinline__ float get3(const char* base, int plane, int row, int r, int c) {
    const char* p = base + r * row + 2 * c;
    const float h = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(p) << 16);
    const float m = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(p + plane) << 16);
    const float l = __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(p + 2 * plane) << 16);
    return (h + m) + l;      // exact
}
// four consecutive features of one sample (an accumulator fragment) into the three planes: one 8-byte write per plane
__device__ __forceinline__ void put3x4(char* base, int plane, int row, int r, int c, const f32x4& v) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3(v[e], h[e], m[e], l[e]);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    char* p = base + r * row + 2 * c;
    *reinterpret_cast<u32x2*>(p) = (u32x2){(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
    *reinterpret_cast<u32x2*>(p + plane) = (u32x2){(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
    *reinterpret_cast<u32x2*>(p + 2 * plane) = (u32x2){(l[0] >> 16) | l[1], (l[2] >> 16) | l[3]};
}
// training: the first n rows (w floats each, w a multiple of 4) of a plane tile back to fp32 rows in global memory
template <int NT, typename AtFn>
__device__ __forceinline__ void save_rows3(float* dst, const char* base, int plane, int row, int w, int n, int tid, AtFn at) {
    const int w4 = w >> 2;
    const float inv = 1.f / (float)w4;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    for (int q = tid; q < n * w4; q += NT) {
        int r, c4;
        row_quad(q, w4, inv, r, c4);
        const char* p = base + r * row + 8 * c4;
        const u32x2 h = *reinterpret_cast<const u32x2*>(p), m = *reinterpret_cast<const u32x2*>(p + plane),
                    l = *reinterpret_cast<const u32x2*>(p + 2 * plane);
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v[2 * i] = (__uint_as_float(h[i] << 16) + __uint_as_float(m[i] << 16)) + __uint_as_float(l[i] << 16);
            v[2 * i + 1] = (__uint_as_float(h[i] & 0xFFFF0000u) + __uint_as_float(m[i] & 0xFFFF0000u)) + __uint_as_float(l[i] & 0xFFFF0000u);
        }
        *reinterpret_cast<f32x4*>(dst + at(r) * (size_t)w + 4 * c4) = v;
    }
}
// acc[j] += W X^T for the wave's feature tile and sample tiles 0 .. NS-1 (NS = 2 or 4) over KS k-steps of 32.  A step is
// one (k-step, tile): its six products alternate between the tile's accumulator (lh, mh, hh) and a scratch accumulator (hl,
// mm, hm — the small terms, added once at the end of the step), so that dependent MFMAs lie two apart.  The order is
// pinned (sched_barrier): left to itself the compiler re-reads operand planes instead of keeping them and waits for every
// read right behind its issue.
template <int KS, int NS>
__device__ __forceinline__ void mma3(const Frag3 (&a)[KS], const char* base, int plane, int row, f32x4 (&acc)[4], int lane) {
    const char* bp = base + (lane & 15) * row + 16 * (lane >> 4);
    constexpr int N = KS * NS;
    // (a step's planes are read at its start: the SIMD's other MLP wave covers the latency.  Reading them one step ahead
    // costs 12 more registers than the kernel has — 24 spilled dwords reloaded inside this loop — and gains nothing: with
    // 6 B per element read by all eight waves the two layers are bound by LDS bandwidth as much as by the matrix pipe,
    // 864 KB per chunk = 6.8 k cycles at 128 B / cycle against 6.9 k cycles of MFMA issue)
#pragma unroll
    for (int st = 0; st < N; ++st) {
        const int ks = st / NS, j = st % NS;
        const char* q = bp + (16 * j) * row + 64 * ks;
        bf16x8 cur[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) cur[p] = *reinterpret_cast<const bf16x8*>(q + p * plane);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks].h, cur[2], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks].l, cur[0], acc[j], 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks].m, cur[1], t, 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks].m, cur[0], acc[j], 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks].h, cur[1], t, 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks].h, cur[0], acc[j], 0, 0, 0);
        acc[j] += t;
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(768) void shade_forward_pipe_kernel(const TfShade S, const TileSrc src, float* __restrict__ rgb_out,
                                                                 const TfShadeSave save) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    constexpr int NC = 512, KTB = PIPE_KB / 16, FC = PIPE_FC, KS1 = PIPE_KX / 32, KS2 = PIPE_FC / 32;
    constexpr int XR = PIPE_XROW, XP = PIPE_XPLANE, HR = PIPE_HROW, HP = PIPE_HPLANE;
    const PipeLds L = pipe_lds();
    int* pre = reinterpret_cast<int*>(ldsb + L.offPre);
    char* w3p = ldsb + L.offW3;                      // planes [3][3][128] bf16
    float* biases = reinterpret_cast<float*>(ldsb + L.offBias);
    shard_prefix(src.counters, src.seg_cap, pre, threadIdx.x);
    for (int i = threadIdx.x; i < 3 * FC; i += 768) put3(w3p, 3 * FC * 2, FC * 2, i / FC, i % FC, S.w3[i]);
    for (int i = threadIdx.x; i < 2 * FC + 3; i += 768) biases[i] = i < FC ? S.b1[i] : (i < 2 * FC ? S.b2[i - FC] : S.b3[i - 2 * FC]);
    __syncthreads();
    const int total = pre[TF_N_SHARDS];
    const long long n_tiles = (total + 15) / 16;
    const int v_begin = (int)(((long long)blockIdx.x * n_tiles) / (long long)gridDim.x) * 16;
    const int v_end = min(total, (int)((((long long)blockIdx.x + 1) * n_tiles) / (long long)gridDim.x) * 16);
    // (a chunk is cut short where it would span a third shard — fwd_locate — so the chunk sequence is walked, not
    // computed.  The gather crew walks it, one chunk per iteration in P2 — a locate is ~2 k cycles of dependent LDS reads —
    // and leaves the descriptors in a ring the MLP crew reads)
    int4_desc* desc = reinterpret_cast<int4_desc*>(ldsb + L.offDesc);
    int tid0 = threadIdx.x;
    asm volatile("" : "+v"(tid0));
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int kp1 = kpad16(S.in_c);
    const int nb = (S.app_dim + 15) >> 4, ktb = kpad16(S.n_app_total) / 16;
    char* Xp = ldsb + L.offX;           // X planes (row stride XR); later the H2 planes (row stride HR, plane stride HP)
    char* H1p = ldsb + L.offH1;
    float* V = reinterpret_cast<float*>(ldsb + L.offV);
    TF_ABL_INIT();
    TF_T0();

    if (wave < 8) {
        // ================= MLP crew =================
        const int f_base = 16 * wave, lane0 = tid0 & 63;
        Frag3 a1[KS1], a2[KS2];
        load_weight_frags3<KS1>(S.w1, kp1, kp1, f_base, lane0, a1);
        load_weight_frags3<KS2>(S.w2, FC, FC, f_base, lane0, a2);
        // basis (fp32): (feature tile, sample tile) pair `wave` (nb <= 2: at most 8 pairs); fetched one phase ahead of its
        // use, every iteration (36 registers the hidden layers need)
        FChunk ck, ck_next;
        ck.s0 = ck.n0 = ck.s1 = ck.n1 = 0;
        bool on = false;
        bool more = fwd_locate(src, pre, v_begin, v_end, ck_next);      // chunk 0: the gather crew's chunk of iteration 0
        TF_MARK(13);
        for (int par = 0, it = 0;; par ^= 1, ++it) {
            // thread coordinates from an opaque copy of the thread id: no per-thread address is computed (and kept in
            // registers) outside the chunk loop — the crew has ~40 registers beside its weights
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63, lc = lane & 15, lg = lane >> 4;
            const int n = on ? ck.n() : 0, nt = (n + 15) >> 4, n16 = 16 * nt;
            const float* ixyz = reinterpret_cast<const float*>(ldsb + (par ? L.offInfo0 : L.offInfo1));
            const float* iview = ixyz + 3 * M;
            TF_MARK(10);
            // ---- P1: the appearance features and view directions of this chunk — computed by the gather crew during the
            // previous iteration's P5 into a side tile (planes of [64][32] in the idle H1 region) — move into the X planes
            if (on && !TF_ABL(16)) {
                for (int q = tid; q < 3 * n16 * 4; q += NC) {
                    const int pl = q / (n16 * 4), rc = q - pl * (n16 * 4);
                    *reinterpret_cast<u32x4*>(Xp + pl * XP + (rc >> 2) * XR + 16 * (rc & 3)) =
                        *reinterpret_cast<const u32x4*>(H1p + pl * (M * 64) + (rc >> 2) * 64 + 16 * (rc & 3));
                }
                if (save.v && (S.n_app_total & 3) == 0)      // training: two thirds of the V rows (the gather crew: the rest)
                    save_rows<768>(save.v, V, L.sv, S.n_app_total, n, tid, [&](int r) { return ck.at(r); });
            }
            TF_MARK(0);
            lds_barrier();
            TF_MARK(1);
            // ---- P2: positional-encoding blocks + zero K padding
            if (on) {
                int off = S.app_dim + 3;
                for (int b = 0; b < (TF_ABL(16) ? 0 : S.n_pe); ++b) {
                    const int src_k = S.pe[b].src, F = S.pe[b].freqs;
                    const int D = src_k == TF_SRC_FEAT ? S.app_dim : 3;
                    const float* mk = S.pe[b].mask;
                    pe_block_put<NC>(off, D, F, mk, tid, [&](int s_, int d) {
                        return src_k == TF_SRC_FEAT ? get3(Xp, XP, XR, s_, d) : (src_k == TF_SRC_VIEW ? iview[s_ * 3 + d] : ixyz[s_ * 3 + d]);
                    }, [&](int s_, int col, float v) { put3(Xp, XP, XR, s_, col, v); }, n16);
                    off += 2 * D * F;
                }
                // zero K padding up to the fixed extent
                const int padw = PIPE_KX - S.in_c;
                for (int i2 = tid; i2 < n16 * padw; i2 += NC) {
                    const int s_ = i2 / padw, c = S.in_c + (i2 - s_ * padw);
                    char* p = Xp + s_ * XR + 2 * c;
                    *reinterpret_cast<unsigned short*>(p) = 0;
                    *reinterpret_cast<unsigned short*>(p + XP) = 0;
                    *reinterpret_cast<unsigned short*>(p + 2 * XP) = 0;
                }
            }
            TF_MARK(2);
            lds_barrier();
            TF_MARK(3);
            // ---- P3: H1 = relu(W1 X + b1)   (training rows leave behind the barrier that completes them, so that the
            // stores drain under the phase's arithmetic)
            if (on) {
                f32x4 acc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (TF_ABL(8)) {}
                else if (nt > 2) mma3<KS1, 4>(a1, Xp, XP, XR, acc, lane);
                else mma3<KS1, 2>(a1, Xp, XP, XR, acc, lane);
                const f32x4 bias = *reinterpret_cast<const f32x4*>(biases + f_base + 4 * lg);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j >= nt) continue;
                    f32x4 h = acc[j] + bias;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = fmaxf(h[e], 0.f);
                    put3x4(H1p, HP, HR, 16 * j + lc, f_base + 4 * lg, h);
                    if (save.h1 && 16 * j + lc < n) *reinterpret_cast<f32x4*>(save.h1 + ck.at(16 * j + lc) * (size_t)FC + f_base + 4 * lg) = h;
                }
            }
            TF_MARK(4);
            lds_barrier();
            TF_MARK(5);
            // ---- P4: H2 = relu(W2 H1 + b2), written over X
            if (on) {
                f32x4 acc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (TF_ABL(8)) {}
                else if (nt > 2) mma3<KS2, 4>(a2, H1p, HP, HR, acc, lane);
                else mma3<KS2, 2>(a2, H1p, HP, HR, acc, lane);
                const f32x4 bias = *reinterpret_cast<const f32x4*>(biases + FC + f_base + 4 * lg);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j >= nt) continue;
                    f32x4 h = acc[j] + bias;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = fmaxf(h[e], 0.f);
                    put3x4(Xp, HP, HR, 16 * j + lc, f_base + 4 * lg, h);
                    if (save.h2 && 16 * j + lc < n) *reinterpret_cast<f32x4*>(save.h2 + ck.at(16 * j + lc) * (size_t)FC + f_base + 4 * lg) = h;
                }
            }
            TF_MARK(6);
            lds_barrier();
            TF_MARK(7);
            // ---- P5: output layer + sigmoid: o[c][s] = sum_f W3[c][f] H2[s][f], W3 as rows 0..2 of a 16-row operand tile
            // (its planes in LDS); wave w < 4 takes sample tile w.   mlp.py:36-38, 66-67
            if (on && wave < nt && !TF_ABL(32)) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
                const char* ap = w3p + min(lc, 2) * (FC * 2) + 16 * lg;
                const char* bp = Xp + (16 * wave + lc) * HR + 16 * lg;
#pragma unroll
                for (int ks = 0; ks < KS2; ++ks) {
                    bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap + 64 * ks), am = *reinterpret_cast<const bf16x8*>(ap + 3 * FC * 2 + 64 * ks),
                           al = *reinterpret_cast<const bf16x8*>(ap + 2 * 3 * FC * 2 + 64 * ks);
                    if (lc >= 3) ah = am = al = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp + 64 * ks), bm = *reinterpret_cast<const bf16x8*>(bp + HP + 64 * ks),
                                 bl = *reinterpret_cast<const bf16x8*>(bp + 2 * HP + 64 * ks);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc2, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc2, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc2, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
                }
                acc += acc2;
                const int row = 16 * wave + lc;      // D[c = 4 lg + reg][s = lc]: lanes 0..15 hold the three channels
                if (lg == 0 && row < n) {
                    float* o = rgb_out + ck.at(row) * 3;
                    o[0] = 1.f / (1.f + expf(-(acc[0] + biases[2 * FC])));
                    o[1] = 1.f / (1.f + expf(-(acc[1] + biases[2 * FC + 1])));
                    o[2] = 1.f / (1.f + expf(-(acc[2] + biases[2 * FC + 2])));
                }
            }
            TF_MARK(8);
            lds_barrier();
            TF_MARK(9);
            if (!more) break;
            ck = ck_next;
            on = true;
            ck_next = desc_chunk(desc[(it + 1) & 3]);      // chunk it + 1, located by the gather crew in this iteration's P2
            more = ck_next.n() > 0;
        }
        TF_PIPE_FLUSH(0, tf_phase_cycles);
    } else {
        // ================= gather crew: appearance rows of the chunk one ahead of the MLP crew's =================
        __builtin_amdgcn_s_setprio(3);      // its few VALU / LDS instructions go ahead of the MLP crew's streams
        const bool quads = pipe_gather_ok(S);
        // Per-sample info runs one more chunk ahead: during the iteration that gathers chunk f, the coordinates (every lane
        // of a sample's 4), ray index and view direction (lane 0 of the 4) of chunk f + 1 are requested in P2 / P3 and
        // written to the info tile in P4 — a whole phase between every request and its use.
        float nx_x[3] = {0.f, 0.f, 0.f};
        auto load_xyz = [&](const FChunk& c, bool have, int smp, int sub, float (&x)[3], int& ray) {
            ray = -1;
#pragma unroll
            for (int a = 0; a < 3; ++a) x[a] = 0.f;
            if (have && smp < c.n()) {
                const size_t s = c.at(smp);
                x[0] = src.app_xyz[s * 3]; x[1] = src.app_xyz[s * 3 + 1]; x[2] = src.app_xyz[s * 3 + 2];
                if (sub == 0 && src.rays) ray = src.app_ray[s];
            }
        };
        auto load_view = [&](int ray, float (&vd)[3]) {
#pragma unroll
            for (int a = 0; a < 3; ++a) vd[a] = 0.f;
            if (ray >= 0) {
                const float* rp = src.rays + (size_t)ray * 6 + 3;
                vd[0] = rp[0]; vd[1] = rp[1]; vd[2] = rp[2];
            }
        };
        auto put_info = [&](float* ixyz, int smp, int n_valid, const float (&x)[3], float (&vd)[3]) {
            if (src.ndc && smp < n_valid) {   // viewdirs / rays_norm  (tensorBase.py:341-343)
                float q = vd[0] * vd[0];
                q = q + vd[1] * vd[1];
                q = q + vd[2] * vd[2];
                const float nrm = sqrtf(q);
                vd[0] = vd[0] / nrm; vd[1] = vd[1] / nrm; vd[2] = vd[2] / nrm;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                ixyz[smp * 3 + a] = x[a];
                ixyz[3 * M + smp * 3 + a] = vd[a];
            }
        };
        FChunk ckf, ckm, ckn;
        ckm.s0 = ckm.n0 = ckm.s1 = ckm.n1 = 0;
        bool on_f = fwd_locate(src, pre, v_begin, v_end, ckf), on_m = false;
        int v = v_begin + ckf.n();
        {      // chunk 0: requested and written here
            const int smp = (tid0 - NC) >> 2, sub = tid0 & 3;
            int ray;
            float vd[3];
            load_xyz(ckf, on_f, smp, sub, nx_x, ray);
            load_view(ray, vd);
            if (on_f && sub == 0) put_info(reinterpret_cast<float*>(ldsb + L.offInfo0), smp, ckf.n(), nx_x, vd);
        }
        // chunk 1: located here, then one chunk per iteration in P5, where this crew is idle (a locate is ~2 k cycles of
        // dependent LDS reads)
        bool on_n = on_f && fwd_locate(src, pre, v, v_end, ckn);
        if (!on_n) ckn.s0 = ckn.n0 = ckn.s1 = ckn.n1 = 0;
        v += ckn.n();
        if (tid0 == NC) desc[1] = (int4_desc){ckn.s0, ckn.n0, ckn.s1, ckn.n1};
        const int fw = wave - 8;
        TF_MARK(13);
        for (int par = 0, it = 0;; par ^= 1, ++it) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int ftid = tid - NC, smp = ftid >> 2, sub = ftid & 3;      // gather: 4 lanes per sample
            const int nf = on_f ? ckf.n() : 0, n16f = 16 * ((nf + 15) >> 4);
            float* info_next = reinterpret_cast<float*>(ldsb + (par ? L.offInfo0 : L.offInfo1));   // (the MLP crew reads it in P1 / P2 only)
            float* vrow = V + smp * L.sv;
            const bool row_on = on_f && smp < n16f && !TF_ABL(4);
            const float u[3] = {nx_x[0], nx_x[1], nx_x[2]};
            TF_MARK(10);
            VmTaps t;
            make_vm_taps(S.grid, u, t);
            __builtin_amdgcn_sched_barrier(0);
            TF_MARK(11);
            // ---- P1: (the MLP crew moves the feature tile) training: the V rows of the MLP crew's chunk leave for the
            // backward, a third of them from here — this crew overwrites the tile from P2 on
            float nn_x[3], nn_v[3];
            int nn_ray;
            auto atm = [&](int r) { return ckm.at(r); };
            if (on_m && save.v) {
                const int nat = S.n_app_total;
                if ((nat & 3) == 0) {
                    save_rows<768>(save.v, V, L.sv, nat, ckm.n(), tid, atm);      // (both crews: 768 threads)
                } else {
                    for (int r = ftid >> 6; r < ckm.n(); r += 4)
                        for (int c = ftid & 63; c < nat; c += 64) save.v[ckm.at(r) * nat + c] = V[r * L.sv + c];
                }
            }
            TF_MARK(0);
            lds_barrier();
            TF_MARK(1);
            // ---- P2: units 0..2; the next chunk's coordinates and ray indices requested
            load_xyz(ckn, on_n, smp, sub, nn_x, nn_ray);
            if (row_on) {
                if (quads) gather_units<0, 3>(S, t, sub, vrow);
                for (int c = S.n_app_total + sub; c < PIPE_KB; c += 4) vrow[c] = 0.f;
            }
            TF_MARK(2);
            lds_barrier();
            TF_MARK(3);
            // ---- P3: units 3..5 (other field shapes: the whole row, tap by tap); the next chunk's view directions requested;
            // training: the MLP crew's X rows (complete behind the barrier, overwritten by H2 in P4) leave from this crew's
            // idle issue slots — reconstructing 10 row quads per thread costs the MLP crew 3 k cycles of its critical phase
            load_view(nn_ray, nn_v);
            if (on_m && save.x) save_rows3<256>(save.x, Xp, XP, XR, kp1, ckm.n(), ftid, atm);
            if (row_on) {
                if (quads) gather_units<3, 3>(S, t, sub, vrow);
                else app_products(S, u, sub, vrow, 4);
            }
            TF_MARK(4);
            lds_barrier();
            TF_MARK(5);
            // ---- P4: units 6..8; the next chunk's info -> LDS; the basis fragments of P5 fetched — and waited for HERE, where
            // this crew has slack (their zero padding is a select per register that the scheduler otherwise sinks to the
            // first MFMA of P5, behind a wait for the whole fetch)
            f32x4 frb[2][KTB][1];
            load_a_frags<1, KTB>(S.basis, 16 * ktb, 0, ktb, tid & 63, frb[0]);
            if (row_on && quads) gather_units<6, 3>(S, t, sub, vrow);
            if (on_n && sub == 0) put_info(info_next, smp, ckn.n(), nn_x, nn_v);
#pragma unroll
            for (int a = 0; a < 3; ++a) nx_x[a] = nn_x[a];
            load_a_frags<1, KTB>(S.basis, 16 * ktb, 16 * (nb - 1), ktb, tid & 63, frb[1]);
#pragma unroll
            for (int kg = 0; kg < KTB; ++kg) {
                if (kg >= ktb) frb[0][kg][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (kg >= ktb || nb < 2) frb[1][kg][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            TF_MARK(6);
            lds_barrier();
            TF_MARK(7);
            TF_MARK(12);
            // ---- P5: (the MLP crew runs the output layer) basis product of the chunk just gathered: feat = B V^T
            // (tensoRF.py:263) and its view directions -> the side tile the MLP crew picks up in its next P1; this wave: sample
            // tile fw, both feature tiles, two accumulator chains each.  Then the chunk after the next is located.
            if (on_f && !TF_ABL(16)) {
                const int lane = tid & 63, lc = lane & 15, lg = lane >> 4;
                if (16 * fw < n16f) {
                    const float* vp = V + (16 * fw + lc) * L.sv + 4 * lg;
                    const float* iview_f = reinterpret_cast<const float*>(ldsb + (par ? L.offInfo1 : L.offInfo0)) + 3 * M;
#pragma unroll
                    for (int bf = 0; bf < 2; ++bf) {
                        f32x4 bv[3];
#pragma unroll
                        for (int kg = 0; kg < 3; ++kg) bv[kg] = *reinterpret_cast<const f32x4*>(vp + 16 * kg);
                        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kg = 0; kg < KTB; ++kg) {
                            const f32x4 b = bv[kg % 3];
                            if (kg + 3 < KTB) bv[kg % 3] = *reinterpret_cast<const f32x4*>(vp + 16 * (kg + 3));
#pragma unroll
                            for (int e = 0; e < 4; e += 2) {
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(frb[bf][kg][0][e], b[e], acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(frb[bf][kg][0][e + 1], b[e + 1], acc1, 0, 0, 0);
                            }
                        }
                        const f32x4 acc = acc0 + acc1;
                        TF_MARK(14);
                        const int row = 16 * fw + lc;
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {      // (branch-free: an unconditional read at a clamped index, two selects)
                            const int f = 16 * bf + 4 * lg + e, d = f - S.app_dim;
                            const float vw = iview_f[row * 3 + min(max(d, 0), 2)];
                            o[e] = d < 0 ? acc[e] : (d < 3 ? vw : 0.f);
                        }
                        put3x4(H1p, M * 64, 64, row, 16 * bf + 4 * lg, o);
                        TF_MARK(15);
                    }
                }
            }
            TF_MARK(11);
            ckm = ckf;
            on_m = on_f;
            ckf = ckn;
            on_f = on_n;
            on_n = on_f && fwd_locate(src, pre, v, v_end, ckn);
            if (!on_n) ckn.s0 = ckn.n0 = ckn.s1 = ckn.n1 = 0;
            v += ckn.n();
            if (ftid == 0) desc[(it + 2) & 3] = (int4_desc){ckn.s0, ckn.n0, ckn.s1, ckn.n1};
            TF_MARK(8);
            lds_barrier();
            TF_MARK(9);
            if (!on_m) break;
        }
        TF_PIPE_FLUSH(512, tf_phase_cycles_w4);
    }
}

typedef void (*shade_fn_t)(const TfShade, const TileSrc, float*, float*, const TfShadeSave);

template <int FT>
shade_fn_t pick_nb(int nb) {
    switch (nb) {
        case 1: return shade_forward_kernel<FT, 1>;
        case 2: return shade_forward_kernel<FT, 2>;
        case 3: return shade_forward_kernel<FT, 3>;
        case 4: return shade_forward_kernel<FT, 4>;
    }
    return nullptr;
}
shade_fn_t pick_kernel(const TfShade& S) {
    const int nb = (S.app_dim + 15) / 16;
    if (S.head != TF_HEAD_MLP) return pick_nb<4>(nb);
    switch (S.feature_c) {
        case 64: return pick_nb<4>(nb);
        case 128: return pick_nb<8>(nb);
        case 256: return pick_nb<16>(nb);
    }
    return nullptr;
}

int launch_shade(const TfShade* S, const TileSrc& src, float* rgb_out, float* feat_out, int blocks, hipStream_t st,
                 TfShadeSave save = TfShadeSave{nullptr, nullptr, nullptr, nullptr}) {
    shade_fn_t fn = pick_kernel(*S);
    if (!fn) return (int)hipErrorInvalidValue;
    if (S->head == TF_HEAD_SH && S->app_dim != 27) return (int)hipErrorInvalidValue;
    const ShadeLds L = shade_lds(*S);
    const size_t bytes = (size_t)L.total * sizeof(float);
    if (bytes > 160 * 1024 - 1024) return (int)hipErrorInvalidValue;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(fn), (size_t)(bytes));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), bytes, st, *S, src, rgb_out, feat_out, save);
    return TF_CHECK_LAUNCH();
}

// 0 = pipelined kernel wherever its shape conditions hold, 1 = the two-workgroups-per-CU kernel everywhere (tests compare
// the two; tf_shade_forward_variant)
int g_forward_variant = 0;

bool pipe_fits(const TfShade& S) {
    return S.head == TF_HEAD_MLP && S.feature_c == PIPE_FC && S.in_c <= PIPE_KX && S.app_dim + 3 <= 32 && S.n_app_total <= PIPE_KB;
}

int launch_shade_pipe(const TfShade* S, const TileSrc& src, float* rgb_out, int blocks, hipStream_t st, const TfShadeSave& save) {
    const size_t bytes = (size_t)pipe_lds().total;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(shade_forward_pipe_kernel), bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(shade_forward_pipe_kernel, dim3(blocks), dim3(768), bytes, st, *S, src, rgb_out, save);
    return TF_CHECK_LAUNCH();
}

}  // namespace

extern "C" {

int tf_shade_forward_variant(int variant) {
    if (variant < 0 || variant > 1) return (int)hipErrorInvalidValue;
    g_forward_variant = variant;
    return 0;
}

int tf_shade_forward(const TfShade* shade, const float* rays, int ndc, const int* counters, int seg_cap,
                     const int* app_ray, const float* app_xyz, float* rgb_out, int max_workgroups,
                     const TfShadeSave* save, tf_stream_t stream) {
    TileSrc src{counters, seg_cap, 0, app_ray, app_xyz, rays, ndc, nullptr, nullptr};
    const int wgs = max_workgroups > 0 && max_workgroups < 512 ? max_workgroups : 512;
    if (save && shade->head != TF_HEAD_MLP && save->x) return (int)hipErrorInvalidValue;   // X rows exist for MLP heads only
    if (g_forward_variant == 0 && pipe_fits(*shade))      // one workgroup per CU: half the slots
        return launch_shade_pipe(shade, src, rgb_out, (wgs + 1) / 2, (hipStream_t)stream,
                                 save ? *save : TfShadeSave{nullptr, nullptr, nullptr, nullptr});
    return launch_shade(shade, src, rgb_out, nullptr, wgs, (hipStream_t)stream, save ? *save : TfShadeSave{nullptr, nullptr, nullptr, nullptr});
}

int tf_appfeature_points(const TfShade* shade, const float* xyz_n, int n, float* out_feat, tf_stream_t stream) {
    if (n <= 0) return 0;
    TileSrc src{nullptr, 0, n, nullptr, xyz_n, nullptr, 0, nullptr, nullptr};
    int blocks = (n + M - 1) / M;
    if (blocks > 512) blocks = 512;
    return launch_shade(shade, src, nullptr, out_feat, blocks, (hipStream_t)stream);
}

int tf_shade_points(const TfShade* shade, const float* pts_n, const float* viewdirs, const float* features, int n,
                    float* rgb_out, tf_stream_t stream) {
    if (n <= 0) return 0;
    if (!pts_n || !viewdirs || !features || !rgb_out) return (int)hipErrorInvalidValue;
    TileSrc src{nullptr, 0, n, nullptr, pts_n, nullptr, 0, viewdirs, features};
    int blocks = (n + M - 1) / M;
    if (blocks > 512) blocks = 512;
    return launch_shade(shade, src, rgb_out, nullptr, blocks, (hipStream_t)stream);
}

#ifdef TF_PHASE_TIMING
int tf_debug_set_flags_fwd(int flags) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tf_dbg_flags), &flags, sizeof(int)); }
int tf_debug_phase_cycles_w4(unsigned long long* out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(tf_phase_cycles_w4), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(tf_phase_cycles_w4), z, sizeof(z));
    }
    return (int)e;
}
int tf_debug_phase_cycles(unsigned long long* out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(tf_phase_cycles), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(tf_phase_cycles), z, sizeof(z));
    }
    return (int)e;
}
#endif

}  // extern "C"
