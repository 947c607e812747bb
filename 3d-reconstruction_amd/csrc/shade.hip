// shade.hip — appearance lookup + shading head on the packed app list.   gfx950, wave64, fp32 MFMA.
//
// A 512-thread workgroup (8 waves) shades one tile of TF_TILE = 64 samples, two workgroups per CU:
//   1. gather: 8 lanes per sample read each bilinear tap of the channel-last appearance planes/lines as
//      contiguous 16-B pieces (128 B per 8-lane group per tap) and write the plane*line products
//      (sum n_comp wide) to an LDS tile V[64][.]                      tensoRF.py:238-260 / :394-410
//   2. basis_mat: feat^T = B . V^T on v_mfma_f32_16x16x4_f32 (exact fp32)  tensoRF.py:263
//   3. MLP input: [feat, view, PE blocks] built in LDS               mlp.py:8-13, 41-66
//   4. 2 hidden layers on the fp32 MFMA (weights streamed from L2, activations in LDS), ReLU fused
//      in the accumulator epilogue; output layer + sigmoid on the VALU   mlp.py:34-38, 66-67
// Workgroups are persistent and walk the tiles of the 64 packed-list shards.
#include "tf_shade.h"

using namespace tf;

namespace {

// training: copy a [n][w] row block of an LDS tile (row stride ld floats, w a multiple of 4) to global rows, 16 B per lane
template <int NT>
__device__ __forceinline__ void save_rows(float* dst, const float* tile, int ld, int w, int n, int tid) {
    const int w4 = w >> 2;
    const float inv = 1.f / (float)w4;
    for (int q = tid; q < n * w4; q += NT) {
        int row, c4;
        row_quad(q, w4, inv, row, c4);
        *reinterpret_cast<f32x4*>(dst + (size_t)q * 4) = *reinterpret_cast<const f32x4*>(tile + row * ld + 4 * c4);
    }
}

// FT = feature_c / 16 hidden feature tiles (4, 8 or 16), NB = ceil(app_dim/16) basis feature tiles.
// 512 threads = 8 waves shade one 64-sample tile; two workgroups share a CU (81,680 B of LDS each), i.e. 4 waves per
// SIMD (<= 128 VGPRs).  A launch has only ~5 tiles per CU: the chain of dependent phases of one tile is spread over
// 8 waves, and the two resident workgroups fill each other's barrier and memory waits.  (Measured: starting half of
// the workgroups half a tile late to break lockstep changed nothing; issuing all taps of a sample at once made the
// gather slower — it is bound by the CU's fetch rate from the Infinity Cache, ~11-18 B/cycle/CU.)
// Wave w: feature tiles NFW*(w % FG) .. +NFW, sample tiles NSW*(w / FG) .. +NSW  (FG = 8 / SG feature groups).
template <int FT, int NB>
__global__ __launch_bounds__(512, 4) void shade_forward_kernel(const TfShade S, const TileSrc src, float* __restrict__ rgb_out,
                                                               float* __restrict__ feat_out, const TfShadeSave save) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // the only LDS object (16-B aligned base)
    constexpr int NT = 512, NW = 8, SG = FT < NW ? NW / FT : 1, FG = NW / SG, NFW = FT / FG, NSW = 4 / SG;
    const ShadeLds L = shade_lds(S);
    int* pre = reinterpret_cast<int*>(lds + L.offPre);
    float* regA = lds + L.offA;   // V, then H1
    float* regB = lds + L.offB;   // X, then H2
    float* ixyz = lds + L.offInfo;           // [64][3]
    float* iview = lds + L.offInfo + 3 * M;  // [64][3]
    const int tid0 = threadIdx.x;

    if (src.counters) {   // tile prefix over shards (every workgroup computes the same table)
        if (tid0 == 0) {
            int run = 0;
            for (int g = 0; g < TF_N_SHARDS; ++g) {
                pre[g] = run;
                run += (src.counters[g * TF_SHARD_STRIDE] + M - 1) / M;
            }
            pre[TF_N_SHARDS] = run;
        }
        __syncthreads();
    }

    TF_T0();
    // Tiles are handed out dynamically (one returning atomic per tile on a per-launch ticket word, fetched one
    // tile ahead): with ~2.5 tiles per resident workgroup a static stride would leave half the chip idle for the
    // third round.  Direct mode (point lists) keeps the static stride.
    int* ticket = src.counters ? const_cast<int*>(src.counters) + TF_TICKET_SLOT : nullptr;
    int* tbox = reinterpret_cast<int*>(lds + L.offPre) + TF_N_SHARDS + 2;
    if (ticket && tid0 == 0) tbox[0] = atomicAdd(ticket, 1);
    __syncthreads();
    for (int t = ticket ? tbox[0] : (int)blockIdx.x;; t = ticket ? tbox[0] : t + (int)gridDim.x) {
        int s0, n;
        if (!locate_tile(src, pre, t, s0, n)) break;
        // thread coordinates from an opaque copy of the thread id, so that no per-thread address of a later phase is
        // computed (and kept in registers) outside the tile loop: the kernel has 128 VGPRs at 4 waves per SIMD
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int wave = tid >> 6, lane = tid & 63;
        __syncthreads();                                   // everyone has read tbox[0]
        // next tile: the returning atomic is issued now and its result parked in a register; it reaches the LDS box
        // at the end of this tile, so thread 0 does not sit on the atomic's latency here
        int t_next = 0;
        if (ticket && tid == 0) t_next = atomicAdd(ticket, 1);
        TF_MARK(7);

        // ---- tile info
        if (tid < M) {
            float x[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
            if (tid < n) {
                const size_t s = (size_t)s0 + tid;
                x[0] = src.app_xyz[s * 3]; x[1] = src.app_xyz[s * 3 + 1]; x[2] = src.app_xyz[s * 3 + 2];
                if (src.rays) {
                    const int ray = src.app_ray[s];
                    const float* rp = src.rays + (size_t)ray * 6 + 3;
                    v[0] = rp[0]; v[1] = rp[1]; v[2] = rp[2];
                    if (src.ndc) {   // viewdirs / rays_norm  (tensorBase.py:341-343)
                        float q = v[0] * v[0];
                        q = q + v[1] * v[1];
                        q = q + v[2] * v[2];
                        const float nrm = sqrtf(q);
                        v[0] = v[0] / nrm; v[1] = v[1] / nrm; v[2] = v[2] / nrm;
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                ixyz[tid * 3 + a] = x[a];
                iview[tid * 3 + a] = v[a];
            }
        }
        __syncthreads();
        TF_MARK(0);

        // ---- 1. appearance gather -> V: 8 lanes per sample
        {
            const int smp = wave * 8 + (lane >> 3), sub = lane & 7;
            float u[3] = {ixyz[smp * 3], ixyz[smp * 3 + 1], ixyz[smp * 3 + 2]};
            float* vrow = regA + smp * L.sv;
            app_products(S, u, sub, vrow, 8);
            for (int c = S.n_app_total + sub; c < kpad16(S.n_app_total); c += 8) vrow[c] = 0.f;
        }
        __syncthreads();
        TF_MARK(1);
        if (save.v) {     // training: the product rows go back to HBM for the backward (dB = dfeat^T V), 16 B per lane
            const int nat = S.n_app_total;
            float* dst = save.v + (size_t)s0 * nat;
            if ((nat & 3) == 0) {
                const int w4 = nat >> 2;
                const float inv = 1.f / (float)w4;
                for (int q = tid; q < n * w4; q += NT) {
                    int row, c4;
                    row_quad(q, w4, inv, row, c4);
                    *reinterpret_cast<f32x4*>(dst + (size_t)q * 4) = *reinterpret_cast<const f32x4*>(regA + row * L.sv + 4 * c4);
                }
            } else {
                for (int smp = wave; smp < n; smp += NW)
                    for (int c = lane; c < nat; c += 64) dst[(size_t)smp * nat + c] = regA[smp * L.sv + c];
            }
        }

        // ---- 2. basis: feat[s][f] = sum_k B[f][k] V[s][k]; (feature tile, sample tile) pairs dealt to the 8 waves
        for (int pr = wave; pr < 4 * NB; pr += NW) {
            const int bf = pr >> 2, bs = pr & 3;
            f32x4 acc[1][1];
            acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mma_block<1, 1>(S.basis, kpad16(S.n_app_total), 16 * bf, regA, L.sv, 16 * bs, kpad16(S.n_app_total) / 16, acc, lane);
            const int smp = 16 * bs + (lane & 15), g = lane >> 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = 16 * bf + 4 * g + e;
                if (f < S.app_dim) {
                    regB[smp * L.sx + f] = acc[0][0][e];
                    if (feat_out && smp < n) feat_out[((size_t)s0 + smp) * S.app_dim + f] = acc[0][0][e];
                }
            }
        }
        if (feat_out) {      // compute_appfeature hook: features only
            if (ticket && tid == 0) tbox[0] = t_next;
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
                rgb_out[((size_t)s0 + smp) * 3 + ch] = o;
            }
            if (ticket && tid == 0) tbox[0] = t_next;
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
                });
                off += 2 * D * F;
            }
            const int kp = kpad16(S.in_c);
            for (int it = tid; it < M * 16; it += NT) {       // the K padding is < 16 columns
                const int smp = it >> 4, c = S.in_c + (it & 15);
                if (c < kp) regB[smp * L.sx + c] = 0.f;
            }
        }
        __syncthreads();
        TF_MARK(3);
        if (save.x) {     // training: the MLP input rows (zero padded to a multiple of 16) for the backward
            const int w4 = kpad16(S.in_c) >> 2;
            const float inv = 1.f / (float)w4;
            float* dst = save.x + (size_t)s0 * (4 * w4);
            for (int q = tid; q < n * w4; q += NT) {
                int row, c4;
                row_quad(q, w4, inv, row, c4);
                *reinterpret_cast<f32x4*>(dst + (size_t)q * 4) = *reinterpret_cast<const f32x4*>(regB + row * L.sx + 4 * c4);
            }
        }

        // ---- 4. hidden layers
        const int FC = S.feature_c;
        const int f_base = 16 * NFW * (wave % FG), s_base = 16 * NSW * (wave / FG);
        {
            f32x4 acc[NFW][NSW];
#pragma unroll
            for (int i = 0; i < NFW; ++i)
#pragma unroll
                for (int j = 0; j < NSW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mma_block<NFW, NSW>(S.w1, kpad16(S.in_c), f_base, regB, L.sx, s_base, kpad16(S.in_c) / 16, acc, lane);
            const int c = lane & 15, g = lane >> 4;
#pragma unroll
            for (int i = 0; i < NFW; ++i) {
                const int f = f_base + 16 * i + 4 * g;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(S.b1 + f);
#pragma unroll
                for (int j = 0; j < NSW; ++j) {
                    f32x4 h = acc[i][j] + bias;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = fmaxf(h[e], 0.f);
                    *reinterpret_cast<f32x4*>(regA + (s_base + 16 * j + c) * L.sh + f) = h;
                }
            }
        }
        __syncthreads();
        TF_MARK(4);
        if (save.h1) save_rows<NT>(save.h1 + (size_t)s0 * FC, regA, L.sh, FC, n, tid);
        {
            f32x4 acc[NFW][NSW];
#pragma unroll
            for (int i = 0; i < NFW; ++i)
#pragma unroll
                for (int j = 0; j < NSW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mma_block<NFW, NSW>(S.w2, kpad16(FC), f_base, regA, L.sh, s_base, FC / 16, acc, lane);
            const int c = lane & 15, g = lane >> 4;
#pragma unroll
            for (int i = 0; i < NFW; ++i) {
                const int f = f_base + 16 * i + 4 * g;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(S.b2 + f);
#pragma unroll
                for (int j = 0; j < NSW; ++j) {
                    f32x4 h = acc[i][j] + bias;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = fmaxf(h[e], 0.f);
                    *reinterpret_cast<f32x4*>(regB + (s_base + 16 * j + c) * L.sh + f) = h;
                }
            }
        }
        __syncthreads();
        TF_MARK(5);
        if (save.h2) save_rows<NT>(save.h2 + (size_t)s0 * FC, regB, L.sh, FC, n, tid);

        // ---- 5. output layer + sigmoid: 8 lanes per sample
        {
            const int smp = tid >> 3, sub = tid & 7;
            const float* h = regB + smp * L.sh;
            float o0 = 0.f, o1 = 0.f, o2 = 0.f;
            for (int f = sub * 4; f < FC; f += 32) {
                const f32x4 hv = *reinterpret_cast<const f32x4*>(h + f);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(S.w3 + f);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(S.w3 + FC + f);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(S.w3 + 2 * FC + f);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0 = fmaf(hv[e], w0[e], o0);
                    o1 = fmaf(hv[e], w1[e], o1);
                    o2 = fmaf(hv[e], w2[e], o2);
                }
            }
            o0 = quad_sum(o0); o1 = quad_sum(o1); o2 = quad_sum(o2);
            o0 += __shfl_xor(o0, 4, 64); o1 += __shfl_xor(o1, 4, 64); o2 += __shfl_xor(o2, 4, 64);
            if (sub == 0 && smp < n) {
                float* o = rgb_out + ((size_t)s0 + smp) * 3;
                o[0] = 1.f / (1.f + expf(-(o0 + S.b3[0])));
                o[1] = 1.f / (1.f + expf(-(o1 + S.b3[1])));
                o[2] = 1.f / (1.f + expf(-(o2 + S.b3[2])));
            }
        }
        if (ticket && tid == 0) tbox[0] = t_next;
        __syncthreads();
        TF_MARK(6);
    }
    TF_FLUSH();
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
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), bytes, st, *S, src, rgb_out, feat_out, save);
    return TF_CHECK_LAUNCH();
}

}  // namespace

extern "C" {

int tf_shade_forward(const TfShade* shade, const float* rays, int ndc, const int* counters, int seg_cap,
                     const int* app_ray, const float* app_xyz, float* rgb_out, int max_workgroups,
                     const TfShadeSave* save, tf_stream_t stream) {
    TileSrc src{counters, seg_cap, 0, app_ray, app_xyz, rays, ndc};
    const int wgs = max_workgroups > 0 && max_workgroups < 512 ? max_workgroups : 512;
    if (save && shade->head != TF_HEAD_MLP && save->x) return (int)hipErrorInvalidValue;   // X rows exist for MLP heads only
    return launch_shade(shade, src, rgb_out, nullptr, wgs, (hipStream_t)stream, save ? *save : TfShadeSave{nullptr, nullptr, nullptr, nullptr});
}

int tf_appfeature_points(const TfShade* shade, const float* xyz_n, int n, float* out_feat, tf_stream_t stream) {
    if (n <= 0) return 0;
    TileSrc src{nullptr, 0, n, nullptr, xyz_n, nullptr, 0};
    int blocks = (n + M - 1) / M;
    if (blocks > 512) blocks = 512;
    return launch_shade(shade, src, nullptr, out_feat, blocks, (hipStream_t)stream);
}

#ifdef TF_PHASE_TIMING
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
