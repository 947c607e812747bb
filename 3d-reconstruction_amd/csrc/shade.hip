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
#include "tf_device.h"

using namespace tf;

namespace {

constexpr int M = TF_TILE;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int kpad16(int k) { return (k + 15) & ~15; }

struct ShadeLds {   // strides in floats; all regions carved from one dynamic LDS array
    int sv, sx, sh;       // row strides of V, X, H
    int offA, offB, offInfo, offPre;
    int total;            // floats
};
__host__ __device__ inline ShadeLds shade_lds(const TfShade& S) {
    ShadeLds L;
    L.sv = kpad16(S.n_app_total) + 4;
    const int xin = S.head == TF_HEAD_MLP ? S.in_c : S.app_dim + 3;
    L.sx = kpad16(xin > S.app_dim + 3 ? xin : S.app_dim + 3) + 4;
    L.sh = (S.head == TF_HEAD_MLP ? S.feature_c : 0) + 4;
    const int a = L.sv > L.sh ? L.sv : L.sh, b = L.sx > L.sh ? L.sx : L.sh;
    L.offA = 0;
    L.offB = M * a;
    L.offInfo = L.offB + M * b;
    L.offPre = L.offInfo + M * 8;           // tile prefix over shards (65 ints)
    L.total = L.offPre + 80;
    return L;
}

// The wave computes D[f][s] += sum_k W[f][k] * X[s][k] for feature tiles f_base+16a (a<NF) and sample
// tiles s_base+16b (b<NS).  W: global, row stride ldw (multiple of 16, zero padded); X: LDS, stride ldx.
// Lane (r = lane&15, kq = lane>>4) loads W[f][16kg+4kq..+3] and X[s][16kg+4kq..+3]; MFMA step e uses
// element e of both, i.e. k = 16kg+4kq+e on both operands.  D: row(feature) = 4*(lane>>4)+reg, col(sample) = lane&15.
template <int NF, int NS>
__device__ __forceinline__ void mma_block(const float* __restrict__ Wg, int ldw, int f_base, const float* Xs, int ldx,
                                          int s_base, int kgroups, f32x4 (&acc)[NF][NS]) {
    const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
    const float* wp = Wg + (size_t)(f_base + r) * ldw + 4 * kq;
    const float* xp = Xs + (s_base + r) * ldx + 4 * kq;
#pragma unroll 2
    for (int kg = 0; kg < kgroups; ++kg) {
        f32x4 a[NF], b[NS];
#pragma unroll
        for (int i = 0; i < NF; ++i) a[i] = *reinterpret_cast<const f32x4*>(wp + (size_t)16 * i * ldw + 16 * kg);
#pragma unroll
        for (int j = 0; j < NS; ++j) b[j] = *reinterpret_cast<const f32x4*>(xp + 16 * j * ldx + 16 * kg);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < NS; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
}

// Products (P*m)(L*m) [VM] or (L0*L1*L2)*m [CP] of one sample, channel quads sub, sub+4, ..., written to
// vrow[0 .. n_app_total).
__device__ __forceinline__ void app_products(const TfShade& S, const float u[3], int sub, float* vrow) {
    if (S.model == TF_MODEL_VM) {
        VmTaps t;
        make_vm_taps(S.grid, u, t);
        int coff = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int C = S.app.n_comp[i];
            const float* mk = S.app.mask[i];
            if ((C & 3) == 0 && (coff & 3) == 0) {
                for (int q = sub; q < (C >> 2); q += 4) {
                    float4_t p = bilerp4(S.app.plane[i], C, t.p[i], q * 4);
                    float4_t l = lerp4(S.app.line[i], C, t.l[i], q * 4);
                    if (mk) {
                        float4_t m = ld4(mk + q * 4);
                        p *= m;
                        l *= m;
                    }
                    *reinterpret_cast<float4_t*>(vrow + coff + q * 4) = p * l;
                }
            } else {
                for (int c = sub; c < C; c += 4) {
                    float p = bilerp1(S.app.plane[i], C, t.p[i], c);
                    float l = lerp1(S.app.line[i], C, t.l[i], c);
                    if (mk) {
                        p *= mk[c];
                        l *= mk[c];
                    }
                    vrow[coff + c] = p * l;
                }
            }
            coff += C;
        }
    } else {
        const int C = S.app.n_comp[0];
        Tap1 t0 = make_tap1(u[vecm(0)], S.grid[vecm(0)]);
        Tap1 t1 = make_tap1(u[vecm(1)], S.grid[vecm(1)]);
        Tap1 t2 = make_tap1(u[vecm(2)], S.grid[vecm(2)]);
        const float* mk = S.app.mask[0];
        if ((C & 3) == 0) {
            for (int q = sub; q < (C >> 2); q += 4) {
                float4_t v = lerp4(S.app.line[0], C, t0, q * 4);
                v *= lerp4(S.app.line[1], C, t1, q * 4);
                v *= lerp4(S.app.line[2], C, t2, q * 4);
                if (mk) v *= ld4(mk + q * 4);
                *reinterpret_cast<float4_t*>(vrow + q * 4) = v;
            }
        } else {
            for (int c = sub; c < C; c += 4) {
                float v = lerp1(S.app.line[0], C, t0, c) * lerp1(S.app.line[1], C, t1, c);
                v *= lerp1(S.app.line[2], C, t2, c);
                if (mk) v *= mk[c];
                vrow[c] = v;
            }
        }
    }
}

// real SH basis, degree 2 (sh.py:87-112)
__device__ __forceinline__ void sh9(const float d[3], float y[9]) {
    const float x = d[0], yy_ = d[1], z = d[2];
    y[0] = 0.28209479177387814f;
    y[1] = -0.4886025119029199f * yy_;
    y[2] = 0.4886025119029199f * z;
    y[3] = -0.4886025119029199f * x;
    const float xx = x * x, yy = yy_ * yy_, zz = z * z;
    y[4] = 1.0925484305920792f * (x * yy_);
    y[5] = -1.0925484305920792f * (yy_ * z);
    y[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
    y[7] = -1.0925484305920792f * (x * z);
    y[8] = 0.5462742152960396f * (xx - yy);
}

struct TileSrc {          // where a tile's samples come from
    const int* counters;  // sharded packed list (NULL in direct mode)
    int seg_cap;
    int n_direct;         // direct mode: a plain point list of this many entries
    const int* app_ray;
    const float* app_xyz;
    const float* rays;
    int ndc;
};

// Enumerates the tiles of all shards: returns false when t is past the last tile.
__device__ __forceinline__ bool locate_tile(const TileSrc& src, const int* pre /*LDS prefix[65]*/, int t, int& s0,
                                            int& n) {
    if (src.counters == nullptr) {
        s0 = t * M;
        n = min(M, src.n_direct - s0);
        return n > 0;
    }
    if (t >= pre[TF_N_SHARDS]) return false;
    int lo = 0, hi = TF_N_SHARDS - 1;   // last g with pre[g] <= t
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pre[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const int lt = t - pre[lo];
    const int cnt = src.counters[lo * TF_SHARD_STRIDE];
    s0 = lo * src.seg_cap + lt * M;
    n = min(M, cnt - lt * M);
    return true;
}

// NF = feature_c / 64 (feature tiles per wave), NB = ceil(app_dim/16) (basis feature tiles).
template <int NF, int NB>
__global__ __launch_bounds__(256) void shade_forward_kernel(const TfShade S, const TileSrc src, float* __restrict__ rgb_out,
                                                            float* __restrict__ feat_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // the only LDS object (16-B aligned base)
    const ShadeLds L = shade_lds(S);
    int* pre = reinterpret_cast<int*>(lds + L.offPre);
    float* regA = lds + L.offA;   // V, then H1
    float* regB = lds + L.offB;   // X, then H2
    int* iray = reinterpret_cast<int*>(lds + L.offInfo);
    float* ixyz = lds + L.offInfo + M;       // [64][3]
    float* iview = lds + L.offInfo + 4 * M;  // [64][3]
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

    for (int t = blockIdx.x;; t += gridDim.x) {
        int s0, n;
        if (!locate_tile(src, pre, t, s0, n)) break;

        // ---- tile info
        if (tid < M) {
            int ray = 0;
            float x[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
            if (tid < n) {
                const size_t s = (size_t)s0 + tid;
                x[0] = src.app_xyz[s * 3]; x[1] = src.app_xyz[s * 3 + 1]; x[2] = src.app_xyz[s * 3 + 2];
                if (src.rays) {
                    ray = src.app_ray[s];
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
            iray[tid] = ray;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                ixyz[tid * 3 + a] = x[a];
                iview[tid * 3 + a] = v[a];
            }
        }
        __syncthreads();

        // ---- 1. appearance gather -> V
        {
            const int smp = wave * 16 + (lane >> 2), sub = lane & 3;
            float u[3] = {ixyz[smp * 3], ixyz[smp * 3 + 1], ixyz[smp * 3 + 2]};
            float* vrow = regA + smp * L.sv;
            app_products(S, u, sub, vrow);
            for (int c = S.n_app_total + sub; c < kpad16(S.n_app_total); c += 4) vrow[c] = 0.f;
        }
        __syncthreads();

        // ---- 2. basis: feat[s][f] = sum_k B[f][k] V[s][k]; wave w owns sample tile w
        {
            f32x4 acc[NB][1];
#pragma unroll
            for (int i = 0; i < NB; ++i) acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mma_block<NB, 1>(S.basis, kpad16(S.n_app_total), 0, regA, L.sv, wave * 16, kpad16(S.n_app_total) / 16, acc);
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
                for (int it = tid; it < M * D; it += 256) {
                    const int smp = it / D, d = it % D;
                    float* x = regB + smp * L.sx;
                    const float v = src_k == TF_SRC_FEAT ? x[d] : (src_k == TF_SRC_VIEW ? iview[smp * 3 + d] : ixyz[smp * 3 + d]);
                    float fr = 1.f;
                    for (int k = 0; k < F; ++k) {
                        const float a = v * fr;
                        float sn, cs;
                        sincosf(a, &sn, &cs);
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

        // ---- 4. hidden layers: wave w owns features [16*NF*w, 16*NF*(w+1)) x all 4 sample tiles
        const int FC = S.feature_c;
        {
            f32x4 acc[NF][4];
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mma_block<NF, 4>(S.w1, kpad16(S.in_c), 16 * NF * wave, regB, L.sx, 0, kpad16(S.in_c) / 16, acc);
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
        {
            f32x4 acc[NF][4];
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mma_block<NF, 4>(S.w2, kpad16(FC), 16 * NF * wave, regA, L.sh, 0, FC / 16, acc);
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
    }
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

}  // extern "C"
