// shade_bwd.hip — backward of the shading head + appearance lookup (autograd of tensoRF.py:230-263 /
// :388-415 and mlp.py:27-155).   gfx950, wave64, fp32 MFMA.
//
// One persistent 256-thread workgroup per CU walks 64-sample tiles of the packed app list.  Per tile it
// recomputes the forward (gather -> V, basis -> feat, PE -> X, two hidden layers -> H1, H2, output) in
// LDS, then back-propagates dL/dc:
//     do  = dL/dc . c(1-c)                         dW3 += do^T H2          db3 += sum do
//     dZ2 = (do W3) . [H2>0]     (in place of H2)  dW2 += dZ2^T H1         db2 += sum dZ2
//     dZ1 = (W2^T dZ2) . [H1>0]  (in place of H1)  dW1 += dZ1^T X          db1 += sum dZ1
//     dX  = W1^T dZ1             (in place of X)   dfeat = dX[:, :D] + PE'(feat) . dX[:, PE cols]
//     dB += dfeat^T V                               dV = B^T dfeat (in place of V)
//     dP / dL scatter-add with float atomics (4 lanes per sample, channel-last gradients).
// All weight-gradient GEMMs accumulate across the workgroup's tiles in MFMA accumulator registers
// (sample index = MFMA k dimension) and are flushed once per workgroup.
#include "tf_shade.h"

using namespace tf;

namespace {

enum { ROW = 0, COL = 1 };

// operand fragment of 4 consecutive k for index `idx`: ROW: p[idx][k..k+3] (k contiguous);
// COL: p[k..k+3][idx] (k strided).
template <int MODE>
__device__ __forceinline__ f32x4 ldfrag(const float* p, int ld, int idx, int k) {
    if (MODE == ROW) return *reinterpret_cast<const f32x4*>(p + (size_t)idx * ld + k);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = p[(size_t)(k + e) * ld + idx];
    return r;
}

// acc[i][j] += sum_k A(a_base+16i + r, k) * B(b_base+16j + r, k), k in [0, 16*kgroups)
template <int NA, int NBT, int AM, int BM>
__device__ __forceinline__ void mma_gen(const float* A, int lda, int a_base, const float* B, int ldb, int b_base,
                                        int kgroups, f32x4 (&acc)[NA][NBT]) {
    const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
#pragma unroll 1
    for (int kg = 0; kg < kgroups; ++kg) {
        const int k = 16 * kg + 4 * kq;
        f32x4 a[NA], b[NBT];
#pragma unroll
        for (int i = 0; i < NA; ++i) a[i] = ldfrag<AM>(A, lda, a_base + 16 * i + r, k);
#pragma unroll
        for (int j = 0; j < NBT; ++j) b[j] = ldfrag<BM>(B, ldb, b_base + 16 * j + r, k);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NBT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
}

template <int NA, int NBT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NA][NBT]) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NBT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

struct BwdLds {
    int sv, sx, sh, sf;
    int offV, offX, offH1, offH2, offDo, offInfo, offPre, total;
};
__host__ __device__ inline BwdLds bwd_lds(const TfShade& S) {
    BwdLds L;
    L.sv = kpad16(S.n_app_total) + 4;
    L.sx = kpad16(S.in_c) + 4;
    L.sh = S.feature_c + 4;
    L.sf = 36;                       // feat copy / dfeat rows (<= 32 used), carved from the H2 region
    L.offV = 0;
    L.offX = L.offV + M * L.sv;
    L.offH1 = L.offX + M * L.sx;
    L.offH2 = L.offH1 + M * L.sh;
    L.offDo = L.offH2 + M * L.sh;
    L.offInfo = L.offDo + M * 4;
    L.offPre = L.offInfo + M * 8;
    L.total = L.offPre + 80;
    return L;
}

// NF = feature_c/64, NB = ceil(app_dim/16) (1..2), KT1 = max k-tiles of the first layer kept in registers.
template <int NF, int NB, int KT1>
__global__ __launch_bounds__(256, 1) void shade_backward_kernel(const TfShade S, const TileSrc src,
                                                                 const float* __restrict__ grad_rgb,
                                                                 const TfShadeGrads G) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const BwdLds L = bwd_lds(S);
    float* V = lds + L.offV;
    float* X = lds + L.offX;
    float* H1 = lds + L.offH1;
    float* H2 = lds + L.offH2;
    float* Fs = H2;                       // feat copy   [64][sf]   (H2 region is free once dZ1 exists)
    float* Fd = H2 + M * L.sf;            // dfeat       [64][sf]
    float* dO = lds + L.offDo;            // [64][4]
    int* iray = reinterpret_cast<int*>(lds + L.offInfo);
    float* ixyz = lds + L.offInfo + M;
    float* iview = lds + L.offInfo + 4 * M;
    int* pre = reinterpret_cast<int*>(lds + L.offPre);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int FC = S.feature_c, kp1 = kpad16(S.in_c), kt1 = kp1 / 16, kpB = kpad16(S.n_app_total), ktB = kpB / 16;
    constexpr int FT = 4 * NF;            // feature tiles of a hidden layer
    constexpr int KTBW = 5;               // basis column tiles per wave kept in registers (n_app <= 320)

    // ---- accumulators that live across tiles
    f32x4 aW2[NF][FT], aW1[NF][KT1], aB[NB][KTBW];
    zero_acc(aW2); zero_acc(aW1); zero_acc(aB);
    float aW3[3] = {0.f, 0.f, 0.f}, ab2 = 0.f, ab1 = 0.f, ab3 = 0.f;

    if (src.counters) {
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
                ray = src.app_ray[s];
                const float* rp = src.rays + (size_t)ray * 6 + 3;
                v[0] = rp[0]; v[1] = rp[1]; v[2] = rp[2];
                if (src.ndc) {
                    float q = v[0] * v[0];
                    q = q + v[1] * v[1];
                    q = q + v[2] * v[2];
                    const float nrm = sqrtf(q);
                    v[0] = v[0] / nrm; v[1] = v[1] / nrm; v[2] = v[2] / nrm;
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

        // ================= forward recompute =================
        {   // gather -> V
            const int smp = wave * 16 + (lane >> 2), sub = lane & 3;
            float u[3] = {ixyz[smp * 3], ixyz[smp * 3 + 1], ixyz[smp * 3 + 2]};
            float* vrow = V + smp * L.sv;
            app_products(S, u, sub, vrow);
            for (int c = S.n_app_total + sub; c < kpB; c += 4) vrow[c] = 0.f;
        }
        __syncthreads();
        {   // basis -> X[:, :app_dim]
            f32x4 acc[NB][1];
            zero_acc(acc);
            mma_block<NB, 1>(S.basis, kpB, 0, V, L.sv, wave * 16, ktB, acc);
            const int smp = wave * 16 + (lane & 15), g = lane >> 4;
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = 16 * i + 4 * g + e;
                    if (f < S.app_dim) X[smp * L.sx + f] = acc[i][0][e];
                }
        }
        if (tid < M) {
#pragma unroll
            for (int a = 0; a < 3; ++a) X[tid * L.sx + S.app_dim + a] = iview[tid * 3 + a];
        }
        __syncthreads();
        {   // PE blocks + zero padding
            int off = S.app_dim + 3;
            for (int b = 0; b < S.n_pe; ++b) {
                const int src_k = S.pe[b].src, F = S.pe[b].freqs;
                const int D = src_k == TF_SRC_FEAT ? S.app_dim : 3;
                const float* mk = S.pe[b].mask;
                for (int it = tid; it < M * D; it += 256) {
                    const int smp = it / D, d = it % D;
                    float* x = X + smp * L.sx;
                    const float v = src_k == TF_SRC_FEAT ? x[d] : (src_k == TF_SRC_VIEW ? iview[smp * 3 + d] : ixyz[smp * 3 + d]);
                    float fr = 1.f;
                    for (int k = 0; k < F; ++k) {
                        float sn, cs;
                        sincosf(v * fr, &sn, &cs);
                        const int ci = d * F + k;
                        if (mk) {
                            sn *= mk[ci];
                            cs *= mk[D * F + ci];
                        }
                        x[off + ci] = sn;
                        x[off + D * F + ci] = cs;
                        fr *= 2.f;
                    }
                }
                off += 2 * D * F;
            }
            for (int it = tid; it < M * (kp1 - S.in_c); it += 256) {
                const int smp = it / (kp1 - S.in_c), c = S.in_c + it % (kp1 - S.in_c);
                X[smp * L.sx + c] = 0.f;
            }
        }
        __syncthreads();
        {   // layer 1 -> H1
            f32x4 acc[NF][4];
            zero_acc(acc);
            mma_block<NF, 4>(S.w1, kp1, 16 * NF * wave, X, L.sx, 0, kt1, acc);
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
                    *reinterpret_cast<f32x4*>(H1 + (16 * j + c) * L.sh + f) = h;
                }
            }
        }
        __syncthreads();
        {   // layer 2 -> H2
            f32x4 acc[NF][4];
            zero_acc(acc);
            mma_block<NF, 4>(S.w2, kpad16(FC), 16 * NF * wave, H1, L.sh, 0, FC / 16, acc);
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
                    *reinterpret_cast<f32x4*>(H2 + (16 * j + c) * L.sh + f) = h;
                }
            }
        }
        __syncthreads();
        {   // output layer, sigmoid, do = dL/dc * c (1 - c)
            const int smp = tid >> 2, sub = tid & 3;
            const float* h = H2 + smp * L.sh;
            float o[3] = {0.f, 0.f, 0.f};
            for (int f = sub * 4; f < FC; f += 16) {
                const f32x4 hv = *reinterpret_cast<const f32x4*>(h + f);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(S.w3 + ch * FC + f);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[ch] = fmaf(hv[e], w[e], o[ch]);
                }
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) o[ch] = quad_sum(o[ch]);
            if (sub < 3) {
                float d = 0.f;
                if (smp < n) {
                    const float c = 1.f / (1.f + expf(-(o[sub] + S.b3[sub])));
                    d = grad_rgb[((size_t)s0 + smp) * 3 + sub] * (c * (1.f - c));
                }
                dO[smp * 4 + sub] = d;
            }
        }
        __syncthreads();
        TF_MARK(0);

        // ================= backward =================
        {   // per-feature pass: dW3, db3, dZ2 (in place of H2), db2.  2 threads per feature (sample halves)
            const int f = tid % FC, half = tid / FC;
            if (half < 2 && FC <= 128) {
                const float w0 = S.w3[f], w1 = S.w3[FC + f], w2 = S.w3[2 * FC + f];
                for (int s = half * 32; s < half * 32 + 32; ++s) {
                    const float d0 = dO[s * 4], d1 = dO[s * 4 + 1], d2 = dO[s * 4 + 2];
                    const float h = H2[s * L.sh + f];
                    aW3[0] = fmaf(d0, h, aW3[0]);
                    aW3[1] = fmaf(d1, h, aW3[1]);
                    aW3[2] = fmaf(d2, h, aW3[2]);
                    const float dz = h > 0.f ? fmaf(d2, w2, fmaf(d1, w1, d0 * w0)) : 0.f;
                    H2[s * L.sh + f] = dz;
                    ab2 += dz;
                }
            }
            if (tid < 3) {
                float a = 0.f;
                for (int s = 0; s < M; ++s) a += dO[s * 4 + tid];
                ab3 += a;
            }
        }
        __syncthreads();
        TF_MARK(1);
        // dW2[f2][f1] += sum_s dZ2[s][f2] H1[s][f1]
        mma_gen<NF, FT, COL, COL>(H2, L.sh, 16 * NF * wave, H1, L.sh, 0, M / 16, aW2);
        __syncthreads();   // every wave is done reading H1 for dW2
        TF_MARK(2);
        {   // dH1[f1][s] = sum_f2 W2[f2][f1] dZ2[s][f2];  dZ1 = dH1 . [H1 > 0] written in place of H1
            f32x4 acc[NF][4];
            zero_acc(acc);
            mma_gen<NF, 4, COL, ROW>(S.w2, kpad16(FC), 16 * NF * wave, H2, L.sh, 0, FC / 16, acc);
            const int c = lane & 15, g = lane >> 4;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int f = 16 * (NF * wave + i) + 4 * g;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float* hp = H1 + (16 * j + c) * L.sh + f;
                    const f32x4 h = *reinterpret_cast<const f32x4*>(hp);
                    f32x4 dz;
#pragma unroll
                    for (int e = 0; e < 4; ++e) dz[e] = h[e] > 0.f ? acc[i][j][e] : 0.f;
                    *reinterpret_cast<f32x4*>(hp) = dz;
                }
            }
        }
        __syncthreads();
        TF_MARK(3);
        {   // db1 += column sums of dZ1
            const int f = tid % FC, half = tid / FC;
            if (half < 2 && FC <= 128) {
                float a = 0.f;
                for (int s = half * 32; s < half * 32 + 32; ++s) a += H1[s * L.sh + f];
                ab1 += a;
            }
        }
        // dW1[f][k] += sum_s dZ1[s][f] X[s][k]: dZ1 fragments are read once per k-group and reused for every
        // k tile of X
        {
            const int r = lane & 15, kq = lane >> 4;
#pragma unroll 1
            for (int kg = 0; kg < M / 16; ++kg) {
                const int ks = 16 * kg + 4 * kq;
                f32x4 a[NF];
#pragma unroll
                for (int i = 0; i < NF; ++i) a[i] = ldfrag<COL>(H1, L.sh, 16 * (NF * wave + i) + r, ks);
#pragma unroll
                for (int j = 0; j < KT1; ++j) {
                    if (j < kt1) {
                        const f32x4 b = ldfrag<COL>(X, L.sx, 16 * j + r, ks);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int i = 0; i < NF; ++i)
                                aW1[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[e], aW1[i][j], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();   // dW1 finished reading X; the H2 region (dZ2) is free
        TF_MARK(4);
        for (int it = tid; it < M * S.app_dim; it += 256) {   // feat copy for the PE derivative
            const int smp = it / S.app_dim, d = it % S.app_dim;
            Fs[smp * L.sf + d] = X[smp * L.sx + d];
        }
        __syncthreads();
        // dX[k][s] = sum_f W1[f][k] dZ1[s][f], written in place of X; wave w owns k tiles w, w+4, ...
        for (int kt = wave; kt < kt1; kt += 4) {
            f32x4 acc[1][4];
            zero_acc(acc);
            mma_gen<1, 4, COL, ROW>(S.w1, kp1, 16 * kt, H1, L.sh, 0, FC / 16, acc);
            const int c = lane & 15, g = lane >> 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(X + (16 * j + c) * L.sx + 16 * kt + 4 * g) = acc[0][j];
        }
        __syncthreads();
        TF_MARK(5);
        {   // dfeat = dX[:, :D] + PE'(feat): d/dx [sin(x 2^k) m_s] = cos(.) 2^k m_s,  d/dx [cos(.) m_c] = -sin(.) 2^k m_c
            for (int it = tid; it < M * 16 * NB; it += 256) {
                const int smp = it / (16 * NB), d = it % (16 * NB);
                float gsum = 0.f;
                if (d < S.app_dim) {
                    const float* dx = X + smp * L.sx;
                    gsum = dx[d];
                    int off = S.app_dim + 3;
                    for (int b = 0; b < S.n_pe; ++b) {
                        const int F = S.pe[b].freqs;
                        const int D = S.pe[b].src == TF_SRC_FEAT ? S.app_dim : 3;
                        if (S.pe[b].src == TF_SRC_FEAT) {
                            const float* mk = S.pe[b].mask;
                            const float v = Fs[smp * L.sf + d];
                            float fr = 1.f;
                            for (int k = 0; k < F; ++k) {
                                float sn, cs;
                                sincosf(v * fr, &sn, &cs);
                                const int ci = d * F + k;
                                const float ms = mk ? mk[ci] : 1.f, mc = mk ? mk[D * F + ci] : 1.f;
                                gsum += dx[off + ci] * (cs * fr * ms);
                                gsum -= dx[off + D * F + ci] * (sn * fr * mc);
                                fr *= 2.f;
                            }
                        }
                        off += 2 * D * F;
                    }
                }
                Fd[smp * L.sf + d] = gsum;
            }
        }
        __syncthreads();
        // dB[f][c] += sum_s dfeat[s][f] V[s][c]; wave w owns column tiles w, w+4, ...
#pragma unroll
        for (int jj = 0; jj < KTBW; ++jj) {
            const int ct = wave + 4 * jj;
            if (ct < ktB) {
                f32x4 one[NB][1];
#pragma unroll
                for (int i = 0; i < NB; ++i) one[i][0] = aB[i][jj];
                mma_gen<NB, 1, COL, COL>(Fd, L.sf, 0, V, L.sv, 16 * ct, M / 16, one);
#pragma unroll
                for (int i = 0; i < NB; ++i) aB[i][jj] = one[i][0];
            }
        }
        __syncthreads();   // dB finished reading V
        // dV[c][s] = sum_f B[f][c] dfeat[s][f], written in place of V
        for (int ct = wave; ct < ktB; ct += 4) {
            f32x4 acc[1][4];
            zero_acc(acc);
            mma_gen<1, 4, COL, ROW>(S.basis, kpB, 16 * ct, Fd, L.sf, 0, NB, acc);
            const int c = lane & 15, g = lane >> 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(V + (16 * j + c) * L.sv + 16 * ct + 4 * g) = acc[0][j];
        }
        __syncthreads();
        TF_MARK(6);
        // scatter-add into the appearance factor gradients
        if (G.dv_out) {
            // binned mode: hand dL/dV to tf_binned_scatter
            for (int it = tid; it < n * S.n_app_total; it += 256) {
                const int smp = it / S.n_app_total, c = it % S.n_app_total;
                G.dv_out[((size_t)s0 + smp) * S.n_app_total + c] = V[smp * L.sv + c];
            }
        } else if (S.model == TF_MODEL_VM) {
            // run-length merged scatter (tf_device.h): each wave takes its 16 samples as one chunk; the X and
            // H1 regions are free by now and hold the per-wave chunk buffers
            const int ctot = S.n_app_total;
            float* cbuf = X + wave * chunk_lds_words(ctot);
            const int ns = min(kChunk, n - wave * 16);
            if (ns > 0) {
                auto u_of = [&](int s_, float* u) {
                    const int smp = wave * 16 + s_;
                    u[0] = ixyz[smp * 3]; u[1] = ixyz[smp * 3 + 1]; u[2] = ixyz[smp * 3 + 2];
                };
                vm_chunk_gather(S.app, S.grid, ctot, ns, u_of, cbuf, lane);
            }
            __syncthreads();
            if (ns > 0) {
                auto dprod = [&](int s_, int c) { return V[(wave * 16 + s_) * L.sv + c]; };
                vm_chunk_merge_scatter(S.app, G.app, S.grid, ctot, ns, dprod, cbuf, lane);
            }
        } else {
            const int nit = pair_iters(S.model, S.app);
            PairVals cur, nxt;
            bool have = false;
            for (int pr = 0; pr < 8; ++pr) {
                const int sA = wave * 16 + 2 * pr, sB = sA + 1;
                if (sA >= n) break;
                const float uA[3] = {ixyz[sA * 3], ixyz[sA * 3 + 1], ixyz[sA * 3 + 2]};
                const float uB[3] = {ixyz[sB * 3], ixyz[sB * 3 + 1], ixyz[sB * 3 + 2]};
                const float* dvA = V + sA * L.sv;
                const float* dvB = V + sB * L.sv;
                auto dprod = [&](int s, int, int c) { return (s ? dvB : dvA)[c]; };
                for (int it = 0; it < nit; ++it) {
                    cp_pair_gather(S.app, G.app, S.grid, uA, uB, true, sB < n, dprod, lane, it, nxt);
                    if (have) pair_commit(cur);
                    cur = nxt;
                    have = true;
                }
            }
            if (have) pair_commit(cur);
        }
        __syncthreads();
        TF_MARK(7);
    }
    TF_FLUSH();

    // ================= flush the workgroup's weight gradients =================
    {
        const int c = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = 16 * (NF * wave + i) + 4 * g + e;
#pragma unroll
                for (int j = 0; j < FT; ++j) atomicAdd(G.w2 + (size_t)f * FC + 16 * j + c, aW2[i][j][e]);
#pragma unroll
                for (int j = 0; j < KT1; ++j) {
                    const int k = 16 * j + c;
                    if (j < kt1 && k < S.in_c) atomicAdd(G.w1 + (size_t)f * S.in_c + k, aW1[i][j][e]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = 16 * i + 4 * g + e;
#pragma unroll
                for (int jj = 0; jj < KTBW; ++jj) {
                    const int cc = 16 * (wave + 4 * jj) + c;
                    if (f < S.app_dim && cc < S.n_app_total) atomicAdd(G.basis + (size_t)f * S.n_app_total + cc, aB[i][jj][e]);
                }
            }
        const int f = tid % FC, half = tid / FC;
        if (half < 2 && FC <= 128) {
            atomicAdd(G.w3 + f, aW3[0]);
            atomicAdd(G.w3 + FC + f, aW3[1]);
            atomicAdd(G.w3 + 2 * FC + f, aW3[2]);
            atomicAdd(G.b2 + f, ab2);
            atomicAdd(G.b1 + f, ab1);
        }
        if (tid < 3) atomicAdd(G.b3 + tid, ab3);
    }
}

typedef void (*bwd_fn_t)(const TfShade, const TileSrc, const float*, const TfShadeGrads);

template <int NF, int NB>
bwd_fn_t pick_kt(int kt1) {
    if (kt1 <= 4) return shade_backward_kernel<NF, NB, 4>;
    if (kt1 <= 8) return shade_backward_kernel<NF, NB, 8>;
    if (kt1 <= 12) return shade_backward_kernel<NF, NB, 12>;
    return nullptr;
}

bwd_fn_t pick_bwd(const TfShade& S) {
    const int nb = (S.app_dim + 15) / 16, kt1 = kpad16(S.in_c) / 16;
    if (S.head != TF_HEAD_MLP || nb > 2 || kpad16(S.n_app_total) / 16 > 20) return nullptr;
    if (S.feature_c == 64) return nb == 1 ? pick_kt<1, 1>(kt1) : pick_kt<1, 2>(kt1);
    if (S.feature_c == 128) return nb == 1 ? pick_kt<2, 1>(kt1) : pick_kt<2, 2>(kt1);
    return nullptr;
}

}  // namespace

extern "C" {

int tf_shade_backward(const TfShade* shade, const float* rays, int ndc, const int* counters, int seg_cap,
                      const int* app_ray, const float* app_xyz, const float* grad_rgb, const TfShadeGrads* grads,
                      tf_stream_t stream) {
    bwd_fn_t fn = pick_bwd(*shade);
    if (!fn) return (int)hipErrorInvalidValue;   // head / width outside the trained configurations
    const BwdLds L = bwd_lds(*shade);
    const size_t bytes = (size_t)L.total * sizeof(float);
    if (bytes > 160 * 1024) return (int)hipErrorInvalidValue;
    if (shade->model == TF_MODEL_VM && 4 * chunk_lds_words(shade->n_app_total) > M * (L.sx + 2 * L.sh))
        return (int)hipErrorInvalidValue;   // scatter chunk buffers live in the X | H1 | H2 regions
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    TileSrc src{counters, seg_cap, 0, app_ray, app_xyz, rays, ndc};
    hipLaunchKernelGGL(fn, dim3(256), dim3(256), bytes, (hipStream_t)stream, *shade, src, grad_rgb, *grads);
    return TF_CHECK_LAUNCH();
}

#ifdef TF_PHASE_TIMING
int tf_debug_set_flags_shade(int flags) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(tf_dbg_flags), &flags, sizeof(int));
}
int tf_debug_phase_cycles_bwd(unsigned long long* out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(tf_phase_cycles), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(tf_phase_cycles), z, sizeof(z));
    }
    return (int)e;
}
#endif

}  // extern "C"
