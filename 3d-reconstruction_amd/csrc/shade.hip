// shade.hip — appearance lookup + shading head on the packed app list.   gfx950, wave64, fp32 MFMA.
//
// A 256-thread workgroup (4 waves) shades one tile of TF_TILE = 64 samples:
//   1. gather: 4 lanes per sample read each bilinear tap of the channel-last appearance planes/lines as
//      contiguous 16-B pieces (64 B per 4-lane group per tap) and write the plane*line products
//      (sum n_comp wide) to an LDS tile V[64][.]                      tensoRF.py:238-260 / :394-410
//   2. basis_mat: feat^T = B . V^T on v_mfma_f32_16x16x4_f32 (exact fp32)  tensoRF.py:263
//   3. MLP input: [feat, view, PE blocks] built in LDS               mlp.py:8-13, 41-66
//   4. 2 hidden layers on the fp32 MFMA (weights streamed from L2, activations in LDS), ReLU fused
//      in the accumulator epilogue; output layer + sigmoid on the VALU   mlp.py:34-38, 66-67
// Workgroups are persistent and walk the tiles of the 64 packed-list shards.
#include "tf_shade.h"

using namespace tf;

namespace {

// NF = feature_c / 64 (feature tiles per wave), NB = ceil(app_dim/16) (basis feature tiles).
template <int NF, int NB>
__global__ __launch_bounds__(256) void shade_forward_kernel(const TfShade S, const TileSrc src, float* __restrict__ rgb_out,
                                                            float* __restrict__ feat_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // the only LDS object (16-B aligned base)
    const ShadeLds L = shade_lds(S);
    int* pre = reinterpret_cast<int*>(lds + L.offPre);
    float* regA = lds + L.offA;   // V, then H1
    float* regB = lds + L.offB;   // X, then H2
    float* ixyz = lds + L.offInfo;           // [64][3]
    float* iview = lds + L.offInfo + 3 * M;  // [64][3]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    if (src.counters) {   // tile prefix over shards (every workgroup computes the same table)
        if (tid == 0) {
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
    if (ticket && tid == 0) tbox[0] = atomicAdd(ticket, 1);
    __syncthreads();
    for (int t = ticket ? tbox[0] : (int)blockIdx.x;; t = ticket ? tbox[0] : t + (int)gridDim.x) {
        int s0, n;
        if (!locate_tile(src, pre, t, s0, n)) break;
        __syncthreads();                                   // everyone has read tbox[0]
        if (ticket && tid == 0) tbox[0] = atomicAdd(ticket, 1);   // next tile, resolved while this one is shaded
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

        // ---- 1. appearance gather -> V
        {
            const int smp = wave * 16 + (lane >> 2), sub = lane & 3;
            float u[3] = {ixyz[smp * 3], ixyz[smp * 3 + 1], ixyz[smp * 3 + 2]};
            float* vrow = regA + smp * L.sv;
            app_products(S, u, sub, vrow);
            for (int c = S.n_app_total + sub; c < kpad16(S.n_app_total); c += 4) vrow[c] = 0.f;
        }
        __syncthreads();
        TF_MARK(1);

        // ---- 2. basis: feat[s][f] = sum_k B[f][k] V[s][k]; wave w owns sample tile w
        {
            f32x4 acc[NB][1];
#pragma unroll
            for (int i = 0; i < NB; ++i) acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mma_block<NB, 1>(S.basis, kpad16(S.n_app_total), 0, regA, L.sv, wave * 16, kpad16(S.n_app_total) / 16, acc, lane);
            const int smp = wave * 16 + (lane & 15), g = lane >> 4;
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = 16 * i + 4 * g + e;
                    if (f < S.app_dim) {
                        regB[smp * L.sx + f] = acc[i][0][e];
                        if (feat_out && smp < n) feat_out[((size_t)s0 + smp) * S.app_dim + f] = acc[i][0][e];
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
                rgb_out[((size_t)s0 + smp) * 3 + ch] = o;
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
                // thread -> (sample, dim) without integer division: dims padded to a power of two
                const int dp = D <= 4 ? 4 : (D <= 32 ? 32 : 64), dsh = D <= 4 ? 2 : (D <= 32 ? 5 : 6);
                for (int it = tid; it < M * dp; it += 256) {
                    const int smp = it >> dsh, d = it & (dp - 1);
                    if (d >= D) continue;
                    float* x = regB + smp * L.sx;
                    const float v = src_k == TF_SRC_FEAT ? x[d] : (src_k == TF_SRC_VIEW ? iview[smp * 3 + d] : ixyz[smp * 3 + d]);
                    float fr = 1.f;
                    for (int k = 0; k < F; ++k) {
                        const float a = v * fr;
                        float sn, cs;
                        pe_sincos(a, &sn, &cs);
                        const int cs_i = d * F + k;
                        if (mk) {
                            sn *= mk[cs_i];
                            cs *= mk[D * F + cs_i];
                        }
                        x[off + cs_i] = sn;
                        x[off + D * F + cs_i] = cs;
                        fr *= 2.f;
                    }
                }
                off += 2 * D * F;
            }
            const int kp = kpad16(S.in_c);
            for (int it = tid; it < M * (kp - S.in_c); it += 256) {
                const int smp = it / (kp - S.in_c), c = S.in_c + it % (kp - S.in_c);
                regB[smp * L.sx + c] = 0.f;
            }
        }
        __syncthreads();
        TF_MARK(3);

        // ---- 4. hidden layers: wave w owns features [16*NF*w, 16*NF*(w+1)) x all 4 sample tiles
        const int FC = S.feature_c;
        {
            f32x4 acc[NF][4];
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mma_block<NF, 4>(S.w1, kpad16(S.in_c), 16 * NF * wave, regB, L.sx, 0, kpad16(S.in_c) / 16, acc, lane);
            const int c = lane & 15, g = lane >> 4;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int f = 16 * (NF * wave + i) + 4 * g;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(S.b1 + f);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 h = acc[i][j] + bias;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = fmaxf(h[e], 0.f);
                    *reinterpret_cast<f32x4*>(regA + (16 * j + c) * L.sh + f) = h;
                }
            }
        }
        __syncthreads();
        TF_MARK(4);
        {
            f32x4 acc[NF][4];
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mma_block<NF, 4>(S.w2, kpad16(FC), 16 * NF * wave, regA, L.sh, 0, FC / 16, acc, lane);
            const int c = lane & 15, g = lane >> 4;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int f = 16 * (NF * wave + i) + 4 * g;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(S.b2 + f);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 h = acc[i][j] + bias;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = fmaxf(h[e], 0.f);
                    *reinterpret_cast<f32x4*>(regB + (16 * j + c) * L.sh + f) = h;
                }
            }
        }
        __syncthreads();
        TF_MARK(5);

        // ---- 5. output layer + sigmoid: 4 lanes per sample
        {
            const int smp = tid >> 2, sub = tid & 3;
            const float* h = regB + smp * L.sh;
            float o0 = 0.f, o1 = 0.f, o2 = 0.f;
            for (int f = sub * 4; f < FC; f += 16) {
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
            if (sub == 0 && smp < n) {
                float* o = rgb_out + ((size_t)s0 + smp) * 3;
                o[0] = 1.f / (1.f + expf(-(o0 + S.b3[0])));
                o[1] = 1.f / (1.f + expf(-(o1 + S.b3[1])));
                o[2] = 1.f / (1.f + expf(-(o2 + S.b3[2])));
            }
        }
        __syncthreads();
        TF_MARK(6);
    }
    TF_FLUSH();
}

typedef void (*shade_fn_t)(const TfShade, const TileSrc, float*, float*);

template <int NF>
shade_fn_t pick_nb(int nb) {
    switch (nb) {
        case 1: return shade_forward_kernel<NF, 1>;
        case 2: return shade_forward_kernel<NF, 2>;
        case 3: return shade_forward_kernel<NF, 3>;
        case 4: return shade_forward_kernel<NF, 4>;
    }
    return nullptr;
}
shade_fn_t pick_kernel(const TfShade& S) {
    const int nb = (S.app_dim + 15) / 16;
    if (S.head != TF_HEAD_MLP) return pick_nb<1>(nb);
    switch (S.feature_c) {
        case 64: return pick_nb<1>(nb);
        case 128: return pick_nb<2>(nb);
        case 256: return pick_nb<4>(nb);
    }
    return nullptr;
}

int launch_shade(const TfShade* S, const TileSrc& src, float* rgb_out, float* feat_out, int blocks, hipStream_t st) {
    shade_fn_t fn = pick_kernel(*S);
    if (!fn) return (int)hipErrorInvalidValue;
    if (S->head == TF_HEAD_SH && S->app_dim != 27) return (int)hipErrorInvalidValue;
    const ShadeLds L = shade_lds(*S);
    const size_t bytes = (size_t)L.total * sizeof(float);
    if (bytes > 160 * 1024 - 1024) return (int)hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), bytes, st, *S, src, rgb_out, feat_out);
    return TF_CHECK_LAUNCH();
}

}  // namespace

extern "C" {

int tf_shade_forward(const TfShade* shade, const float* rays, int ndc, const int* counters, int seg_cap,
                     const int* app_ray, const float* app_xyz, float* rgb_out, tf_stream_t stream) {
    TileSrc src{counters, seg_cap, 0, app_ray, app_xyz, rays, ndc};
    return launch_shade(shade, src, rgb_out, nullptr, 512, (hipStream_t)stream);
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
