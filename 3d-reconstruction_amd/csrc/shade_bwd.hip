// shade_bwd.hip — backward of the shading head + appearance lookup (autograd of tensoRF.py:230-263 /
// :388-415 and mlp.py:27-155).   gfx950, wave64, fp32 MFMA.
//
// One persistent 512-thread workgroup (8 waves, 2 per SIMD) per CU walks 64-sample tiles of the packed app list.  Per tile it
// recomputes the forward (gather -> V, basis -> feat, PE -> X, two hidden layers -> H1, H2, output) in
// LDS, then back-propagates dL/dc:
//     do  = dL/dc . c(1-c)                         dW3 += do^T H2          db3 += sum do
//     dZ2 = (do W3) . [H2>0]     (in place of H2)  dW2 += dZ2^T H1         db2 += sum dZ2
//     dZ1 = (W2^T dZ2) . [H1>0]  (in place of H1)  dW1 += dZ1^T X          db1 += sum dZ1
//     dX  = W1^T dZ1             (in place of X)   dfeat = dX[:, :D] + PE'(feat) . dX[:, PE cols]
//     dB += dfeat^T V                               dV = B^T dfeat (in place of V)
//     dP / dL scatter-add with float atomics (4 lanes per sample, channel-last gradients).
// The weight-gradient GEMMs (sample index = MFMA k dimension) accumulate in REGISTERS across the tiles of a
// workgroup (104 VGPRs per lane at 2 waves per SIMD, no scratch: kernels that spill cannot be replayed from a
// hipGraph on this stack) and are written once, in accumulator-fragment order, to the workgroup's slab in global
// memory; wslab_reduce_kernel sums the slabs over the workgroups.
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
                                        int kgroups, f32x4 (&acc)[NA][NBT], int lane) {
    const int r = lane & 15, kq = lane >> 4;
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
    int offV, offX, offH1, offH2, offF, offDo, offInfo, offPre, total;
    int rg;      // 1: V does not fit next to X, H1, H2 -> it shares the H1 | H2 space and is gathered a second time
};
__host__ __device__ inline BwdLds bwd_lds(const TfShade& S) {
    BwdLds L;
    L.sv = kpad16(S.n_app_total) + 4;
    L.sx = kpad16(S.in_c) + 4;
    L.sh = S.feature_c + 4;
    L.sf = 36;                       // feat copy / dfeat rows (<= 32 used)
    L.rg = 0;
    L.offV = 0;
    L.offX = L.offV + M * L.sv;
    L.offH1 = L.offX + M * L.sx;
    L.offH2 = L.offH1 + M * L.sh;
    L.offF = L.offH2;                // carved from the H2 region, which is free once dZ1 exists
    L.offDo = L.offH2 + M * L.sh;
    if ((size_t)(L.offDo + M * 12 + 80) * sizeof(float) > 160 * 1024) {
        // Wide appearance bases (e.g. TensorCP [96]/[288] at featureC 128): V is only needed before the MLP (basis ->
        // feat) and after it (dB, dV), so it takes the H1 | H2 space (plus what it needs beyond) and the kernel
        // gathers it again once dX is done; feat copy / dfeat rows move out of its way.
        L.rg = 1;
        L.offX = 0;
        L.offH1 = L.offX + M * L.sx;
        L.offH2 = L.offH1 + M * L.sh;
        L.offV = L.offH1;
        const int end_v = L.offV + M * L.sv, end_h = L.offH2 + M * L.sh;
        L.offF = end_v > end_h ? end_v : end_h;
        L.offDo = L.offF + 2 * M * L.sf;
    }
    L.offInfo = L.offDo + M * 4;
    L.offPre = L.offInfo + M * 8;
    L.total = L.offPre + 80;
    return L;
}

// floats of one workgroup's weight-gradient slab: dW2 | dW1 | dB tiles of 256 floats each
__host__ __device__ inline size_t wslab_floats(const TfShade& S) {
    const int FT = S.feature_c / 16, kt1 = kpad16(S.in_c) / 16, NB = (S.app_dim + 15) / 16, ktB = kpad16(S.n_app_total) / 16;
    return (size_t)(FT * FT + FT * kt1 + NB * ktB) * 256;
}

// The packed sample list is cut EVENLY over the persistent workgroups: workgroup w owns samples [w q, (w+1) q) of the
// shards' concatenation, q = max(64, ceil(S / workgroups)), and walks them in chunks of <= 64.  (Whole 64-sample tiles
// dealt round-robin leave most workgroups one tile short of the busiest: at config 2, 1281-1288 tiles on 256
// workgroups are 6 rounds instead of 5.03.)  A chunk may straddle a shard boundary, so it has two pieces.
struct Chunk {
    int s0, n0, s1, n1;      // packed positions / lengths of the two pieces (n1 may be 0)
    __device__ __forceinline__ int n() const { return n0 + n1; }
    __device__ __forceinline__ size_t at(int k) const { return k < n0 ? (size_t)s0 + k : (size_t)s1 + (k - n0); }
};
__host__ __device__ inline int samples_per_wg(int total, int n_wg) {
    const int q = (total + n_wg - 1) / n_wg;
    return q < M ? M : q;
}
// spre: exclusive prefix of the shards' sample counts (TF_N_SHARDS + 1 entries); v in [0, v_end)
__device__ __forceinline__ bool locate_chunk(const TileSrc& src, const int* spre, int v, int v_end, Chunk& c) {
    c.s0 = c.n0 = c.s1 = c.n1 = 0;
    if (v >= v_end) return false;
    int g = 0;
#pragma unroll
    for (int k = 1; k < TF_N_SHARDS; ++k) g = spre[k] <= v ? k : g;      // last shard starting at or before v
    const int want = min(M, v_end - v);
    c.s0 = g * src.seg_cap + (v - spre[g]);
    c.n0 = min(want, spre[g + 1] - v);
    if (c.n0 < want) {
        int g2 = g + 1;
        while (g2 < TF_N_SHARDS && spre[g2 + 1] == spre[g2]) ++g2;       // skip empty shards
        if (g2 < TF_N_SHARDS) {
            c.s1 = g2 * src.seg_cap;
            c.n1 = min(want - c.n0, spre[g2 + 1] - spre[g2]);
        }
    }
    return true;
}

// FT = feature_c/16 hidden feature tiles (4 or 8), NB = ceil(app_dim/16) (1..2), KT1 = upper bound of the first
// layer's k tiles kept in registers.  512 threads = 8 waves = 2 per SIMD (256 registers each, no scratch:
// kernels that spill cannot be replayed from a hipGraph on this stack).
// Wave w: feature tile ft = w % FT, sample group sg = w / FT (SG = 8/FT groups of NSW = 4/SG sample tiles).
template <int FT, int NB, int KT1, bool RG>
__global__ __launch_bounds__(512, 2) void shade_backward_kernel(const TfShade S, const TileSrc src,
                                                                 const float* __restrict__ grad_rgb,
                                                                 const TfShadeGrads G) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = 512, NW = 8, SG = NW / FT, NSW = 4 / SG, NW2 = FT / SG, KT1S = KT1 / SG;
    const BwdLds L = bwd_lds(S);
    float* V = lds + L.offV;
    float* X = lds + L.offX;
    float* H1 = lds + L.offH1;
    float* H2 = lds + L.offH2;
    float* Fs = lds + L.offF;             // feat copy   [64][sf]
    float* Fd = Fs + M * L.sf;            // dfeat       [64][sf]
    float* dO = lds + L.offDo;            // [64][4]
    int* iray = reinterpret_cast<int*>(lds + L.offInfo);
    float* ixyz = lds + L.offInfo + M;
    float* iview = lds + L.offInfo + 4 * M;
    int* pre = reinterpret_cast<int*>(lds + L.offPre);
    const int tid = threadIdx.x;
    const int FC = S.feature_c, kp1 = kpad16(S.in_c), kt1 = kp1 / 16, kpB = kpad16(S.n_app_total), ktB = kpB / 16;

    // ---- weight-gradient slab of this workgroup (fragment order: [tile][lane][4]) and the few scalars that
    // do live in registers across tiles
    float* slabW2 = G.wslab + (size_t)blockIdx.x * wslab_floats(S);
    float* slabW1 = slabW2 + FT * FT * 256;
    float* slabB = slabW1 + FT * kt1 * 256;
    bool first = true;
    float aW3[3] = {0.f, 0.f, 0.f}, ab2 = 0.f, ab1 = 0.f, ab3 = 0.f;
    // The weight-gradient GEMMs (dW2, dW1, dB: sample index = MFMA k dimension) accumulate in registers across the
    // tiles of this workgroup — 104 VGPRs per lane — and reach the workgroup's slab once, at the end.  (Variants
    // measured: all three through per-tile slab round trips 411 us, dW1 alone through the slab 402 us, none 392 us;
    // fetching a phase's weight fragments ahead of its MFMA loop on top of this does not fit the register file.)
    f32x4 aW2[1][NW2];
    zero_acc(aW2);
    f32x4 aW1[KT1S];                 // dW1: k tiles my_sg, my_sg + SG, ... of feature tile my_ft
#pragma unroll
    for (int q = 0; q < KT1S; ++q) aW1[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int KTBW = 3;          // dB: column tiles wave, wave + 8, wave + 16 (n_app_total <= 384)
    f32x4 aB[KTBW][NB][1];
#pragma unroll
    for (int k = 0; k < KTBW; ++k)
#pragma unroll
        for (int i = 0; i < NB; ++i) aB[k][i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (tid == 0) {
        int run = 0;
        for (int g = 0; g < TF_N_SHARDS; ++g) {
            pre[g] = run;
            run += src.counters[g * TF_SHARD_STRIDE];
        }
        pre[TF_N_SHARDS] = run;
    }
    __syncthreads();
    const int q_wg = samples_per_wg(pre[TF_N_SHARDS], (int)gridDim.x);
    const int v_end = min(pre[TF_N_SHARDS], ((int)blockIdx.x + 1) * q_wg);

    // per-sample tile info of thread tid < 64, loaded for the NEXT chunk while the current one is processed
    int nx_ray = 0;
    float nx_x[3] = {0.f, 0.f, 0.f}, nx_v[3] = {0.f, 0.f, 0.f};
    auto fetch_info = [&](const Chunk& ck, int tid) {   // tid passed in: the tile loop hands its opaque copy
        nx_ray = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) nx_x[a] = nx_v[a] = 0.f;
        if (tid < ck.n()) {
            const size_t s = ck.at(tid);
            nx_x[0] = src.app_xyz[s * 3]; nx_x[1] = src.app_xyz[s * 3 + 1]; nx_x[2] = src.app_xyz[s * 3 + 2];
            nx_ray = src.app_ray[s];
            const float* rp = src.rays + (size_t)nx_ray * 6 + 3;
            nx_v[0] = rp[0]; nx_v[1] = rp[1]; nx_v[2] = rp[2];
        }
    };
    {
        Chunk c1;
        if (tid < M && locate_chunk(src, pre, (int)blockIdx.x * q_wg, v_end, c1)) fetch_info(c1, tid);
    }

    TF_T0();
    for (int v = (int)blockIdx.x * q_wg;;) {
        Chunk ck;
        if (!locate_chunk(src, pre, v, v_end, ck)) break;
        const int n = ck.n();
        v += n;
        // Thread coordinates are re-derived per tile from an opaque copy of the thread id: otherwise the compiler
        // hoists every phase's per-thread addresses out of the tile loop and runs out of registers (scratch
        // spills, which also break hipGraph replay on this stack).
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int wave = tid >> 6, lane = tid & 63;
        const int my_ft = wave % FT, my_sg = wave / FT, s_base = my_sg * NSW * 16;
        const int lc = lane & 15, lg = lane >> 4;

        // ---- tile info (fetched one tile ahead: app_ray -> rays is a chain of two global latencies)
        if (tid < M) {
            if (src.ndc && tid < n) {   // viewdirs / rays_norm  (tensorBase.py:341-343)
                float q = nx_v[0] * nx_v[0];
                q = q + nx_v[1] * nx_v[1];
                q = q + nx_v[2] * nx_v[2];
                const float nrm = sqrtf(q);
                nx_v[0] = nx_v[0] / nrm; nx_v[1] = nx_v[1] / nrm; nx_v[2] = nx_v[2] / nrm;
            }
            iray[tid] = nx_ray;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                ixyz[tid * 3 + a] = nx_x[a];
                iview[tid * 3 + a] = nx_v[a];
            }
        }
        lds_barrier();
        {
            Chunk c1;
            if (tid < M && locate_chunk(src, pre, v, v_end, c1)) fetch_info(c1, tid);
        }

        // ================= forward recompute =================
        {   // gather -> V, 8 lanes per sample
            const int smp = wave * 8 + (lane >> 3), sub = lane & 7;
            float u[3] = {ixyz[smp * 3], ixyz[smp * 3 + 1], ixyz[smp * 3 + 2]};
            float* vrow = V + smp * L.sv;
            app_products(S, u, sub, vrow, 8);
            for (int c = S.n_app_total + sub; c < kpB; c += 8) vrow[c] = 0.f;
        }
        lds_barrier();
        TF_MARK(8);
        if (wave < 4 * NB) {   // basis -> X[:, :app_dim]: one (feature tile, sample tile) per wave
            const int bf = wave >> 2, bs = wave & 3;
            f32x4 acc[1][1];
            zero_acc(acc);
            mma_block<1, 1>(S.basis, kpB, 16 * bf, V, L.sv, 16 * bs, ktB, acc, lane);
            const int smp = 16 * bs + lc;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = 16 * bf + 4 * lg + e;
                if (f < S.app_dim) X[smp * L.sx + f] = acc[0][0][e];
            }
        }
        if (tid < M) {
#pragma unroll
            for (int a = 0; a < 3; ++a) X[tid * L.sx + S.app_dim + a] = iview[tid * 3 + a];
        }
        lds_barrier();
        TF_MARK(9);
        {   // PE blocks + zero padding
            int off = S.app_dim + 3;
            for (int b = 0; b < S.n_pe; ++b) {
                const int src_k = S.pe[b].src, F = S.pe[b].freqs;
                const int D = src_k == TF_SRC_FEAT ? S.app_dim : 3;
                const float* mk = S.pe[b].mask;
                const int sx = L.sx;
                pe_block<NT>(X, L.sx, off, D, F, mk, tid, [&](int smp, int d) {
                    return src_k == TF_SRC_FEAT ? X[smp * sx + d] : (src_k == TF_SRC_VIEW ? iview[smp * 3 + d] : ixyz[smp * 3 + d]);
                });
                off += 2 * D * F;
            }
            for (int it = tid; it < M * 16; it += NT) {        // the K padding is < 16 columns
                const int smp = it >> 4, c = S.in_c + (it & 15);
                if (c < kp1) X[smp * L.sx + c] = 0.f;
            }
        }
        lds_barrier();
        TF_MARK(10);
        {   // layer 1 -> H1
            f32x4 acc[1][NSW];
            zero_acc(acc);
            mma_block<1, NSW>(S.w1, kp1, 16 * my_ft, X, L.sx, s_base, kt1, acc, lane);
            const int f = 16 * my_ft + 4 * lg;
            const f32x4 bias = *reinterpret_cast<const f32x4*>(S.b1 + f);
#pragma unroll
            for (int j = 0; j < NSW; ++j) {
                f32x4 h = acc[0][j] + bias;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = fmaxf(h[e], 0.f);
                *reinterpret_cast<f32x4*>(H1 + (s_base + 16 * j + lc) * L.sh + f) = h;
            }
        }
        lds_barrier();
        TF_MARK(11);
        {   // layer 2 -> H2
            f32x4 acc[1][NSW];
            zero_acc(acc);
            mma_block<1, NSW>(S.w2, kpad16(FC), 16 * my_ft, H1, L.sh, s_base, FC / 16, acc, lane);
            const int f = 16 * my_ft + 4 * lg;
            const f32x4 bias = *reinterpret_cast<const f32x4*>(S.b2 + f);
#pragma unroll
            for (int j = 0; j < NSW; ++j) {
                f32x4 h = acc[0][j] + bias;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = fmaxf(h[e], 0.f);
                *reinterpret_cast<f32x4*>(H2 + (s_base + 16 * j + lc) * L.sh + f) = h;
            }
        }
        lds_barrier();
        TF_MARK(12);
        {   // output layer, sigmoid, do = dL/dc * c (1 - c); 8 lanes per sample
            const int smp = tid >> 3, sub = tid & 7;
            const float* h = H2 + smp * L.sh;
            float o[3] = {0.f, 0.f, 0.f};
            for (int f = sub * 4; f < FC; f += 32) {
                const f32x4 hv = *reinterpret_cast<const f32x4*>(h + f);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(S.w3 + ch * FC + f);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[ch] = fmaf(hv[e], w[e], o[ch]);
                }
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                o[ch] = quad_sum(o[ch]);
                o[ch] += __shfl_xor(o[ch], 4, 64);
            }
            if (sub < 3) {
                const float ov = sub == 0 ? o[0] : (sub == 1 ? o[1] : o[2]);
                float d = 0.f;
                if (smp < n) {
                    const float c = 1.f / (1.f + expf(-(ov + S.b3[sub])));
                    d = grad_rgb[ck.at(smp) * 3 + sub] * (c * (1.f - c));
                }
                dO[smp * 4 + sub] = d;
            }
        }
        lds_barrier();
        TF_MARK(0);

        // ================= backward =================
        const int pf = tid % FC, pslice = tid / FC, pspan = M / (NT / FC);   // per-feature passes: thread -> (f, sample slice)
        {   // dW3, db3, dZ2 (in place of H2), db2
            const float w0 = S.w3[pf], w1 = S.w3[FC + pf], w2 = S.w3[2 * FC + pf];
            for (int s = pslice * pspan; s < (pslice + 1) * pspan; ++s) {
                const float d0 = dO[s * 4], d1 = dO[s * 4 + 1], d2 = dO[s * 4 + 2];
                const float h = H2[s * L.sh + pf];
                aW3[0] = fmaf(d0, h, aW3[0]);
                aW3[1] = fmaf(d1, h, aW3[1]);
                aW3[2] = fmaf(d2, h, aW3[2]);
                const float dz = h > 0.f ? fmaf(d2, w2, fmaf(d1, w1, d0 * w0)) : 0.f;
                H2[s * L.sh + pf] = dz;
                ab2 += dz;
            }
            if (tid < 3) {
                float a = 0.f;
                for (int s = 0; s < M; ++s) a += dO[s * 4 + tid];
                ab3 += a;
            }
        }
        lds_barrier();
        TF_MARK(1);
        {   // dW2[f2][f1] += sum_s dZ2[s][f2] H1[s][f1]
            mma_gen<1, NW2, COL, COL>(H2, L.sh, 16 * my_ft, H1, L.sh, 16 * NW2 * my_sg, M / 16, aW2, lane);
        }
        lds_barrier();   // every wave is done reading H1 for dW2
        TF_MARK(2);
        {   // dH1[f1][s] = sum_f2 W2[f2][f1] dZ2[s][f2];  dZ1 = dH1 . [H1 > 0] written in place of H1
            f32x4 acc[1][NSW];
            zero_acc(acc);
            mma_block<1, NSW>(S.w2t, FC, 16 * my_ft, H2, L.sh, s_base, FC / 16, acc, lane);
            const int f = 16 * my_ft + 4 * lg;
#pragma unroll
            for (int j = 0; j < NSW; ++j) {
                float* hp = H1 + (s_base + 16 * j + lc) * L.sh + f;
                const f32x4 h = *reinterpret_cast<const f32x4*>(hp);
                f32x4 dz;
#pragma unroll
                for (int e = 0; e < 4; ++e) dz[e] = h[e] > 0.f ? acc[0][j][e] : 0.f;
                *reinterpret_cast<f32x4*>(hp) = dz;
            }
        }
        lds_barrier();
        TF_MARK(3);
        {   // db1 += column sums of dZ1
            float a = 0.f;
            for (int s = pslice * pspan; s < (pslice + 1) * pspan; ++s) a += H1[s * L.sh + pf];
            ab1 += a;
        }
        // dW1[f][k] += sum_s dZ1[s][f] X[s][k]: dZ1 fragments are read once per k-group and reused for every k tile
        // this wave owns (k tiles my_sg, my_sg + SG, ...); the accumulators live in registers across tiles
#pragma unroll 1
        for (int kg = 0; kg < M / 16; ++kg) {
            const int ks = 16 * kg + 4 * lg;
            const f32x4 a = ldfrag<COL>(H1, L.sh, 16 * my_ft + lc, ks);
            // GQ k tiles at a time: their MFMA chains interleave (a chain of four dependent MFMAs per tile would wait
            // on its own accumulator)
            constexpr int GQ = KT1S % 4 == 0 ? 4 : 2;
            static_assert(KT1S % GQ == 0, "k tiles per wave come in groups");
#pragma unroll
            for (int q0 = 0; q0 < KT1S; q0 += GQ) {
                f32x4 b[GQ];
#pragma unroll
                for (int g = 0; g < GQ; ++g) {
                    const int j = my_sg + SG * (q0 + g);
                    b[g] = j < kt1 ? ldfrag<COL>(X, L.sx, 16 * j + lc, ks) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < GQ; ++g)
                        if (my_sg + SG * (q0 + g) < kt1)
                            aW1[q0 + g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[g][e], aW1[q0 + g], 0, 0, 0);
            }
        }
        lds_barrier();   // dW1 finished reading X; the H2 region (dZ2) is free
        TF_MARK(4);
        for (int smp = wave; smp < M; smp += NW)             // feat copy for the PE derivative
            if (lane < S.app_dim) Fs[smp * L.sf + lane] = X[smp * L.sx + lane];
        lds_barrier();
        // dX[k][s] = sum_f W1[f][k] dZ1[s][f], written in place of X.  Work items are (k tile, pair of sample tiles):
        // 2 kt1 items dealt round-robin, so that e.g. 10 k tiles load the 8 waves evenly (whole k tiles would give two
        // waves twice the work)
        for (int it = wave; it < 2 * kt1; it += NW) {
            const int kt = it >> 1, sp = it & 1;
            f32x4 acc[1][2];
            zero_acc(acc);
            mma_block<1, 2>(S.w1t, FC, 16 * kt, H1, L.sh, 32 * sp, FC / 16, acc, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4*>(X + (32 * sp + 16 * j + lc) * L.sx + 16 * kt + 4 * lg) = acc[0][j];
        }
        lds_barrier();
        TF_MARK(5);
        if (RG) {   // V again (its space held H1 / H2 meanwhile); lands while the PE derivative below computes
            const int smp = wave * 8 + (lane >> 3), sub = lane & 7;
            float u[3] = {ixyz[smp * 3], ixyz[smp * 3 + 1], ixyz[smp * 3 + 2]};
            float* vrow = V + smp * L.sv;
            app_products(S, u, sub, vrow, 8);
            for (int c = S.n_app_total + sub; c < kpB; c += 8) vrow[c] = 0.f;
        }
        {   // dfeat = dX[:, :D] + PE'(feat): d/dx [sin(x 2^k) m_s] = cos(.) 2^k m_s,  d/dx [cos(.) m_c] = -sin(.) 2^k m_c
            for (int it = tid; it < M * 16 * NB; it += NT) {
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
                            const bool big = !(ldexpf(fabsf(v), F - 1) < 8192.f);
                            for (int k = 0; k < F; ++k) {
                                float sn, cs;
                                if (__builtin_expect(big, 0)) pe_sincos(v * fr, &sn, &cs);
                                else pe_sincos_fast(v * fr, &sn, &cs);
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
        lds_barrier();
        // dB[f][c] += sum_s dfeat[s][f] V[s][c]; wave w owns column tiles w, w+8, w+16 (register accumulators)
#pragma unroll
        for (int k = 0; k < KTBW; ++k) {
            const int ct = wave + NW * k;
            if (ct < ktB) mma_gen<NB, 1, COL, COL>(Fd, L.sf, 0, V, L.sv, 16 * ct, M / 16, aB[k], lane);
        }
        lds_barrier();   // dB finished reading V
        // dV[c][s] = sum_f B[f][c] dfeat[s][f], written in place of V
        for (int ct = wave; ct < ktB; ct += NW) {
            f32x4 acc[1][4];
            zero_acc(acc);
            mma_gen<1, 4, COL, ROW>(S.basis, kpB, 16 * ct, Fd, L.sf, 0, NB, acc, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(V + (16 * j + lc) * L.sv + 16 * ct + 4 * lg) = acc[0][j];
        }
        lds_barrier();
        TF_MARK(6);
        // hand dL/dV to the scatter stage (tf_binned_scatter for VM, app_direct_scatter_kernel otherwise)
        for (int smp = wave; smp < n; smp += NW)
            for (int c = lane; c < S.n_app_total; c += 64)
                G.dv_out[ck.at(smp) * S.n_app_total + c] = V[smp * L.sv + c];
        first = false;
        lds_barrier();
        TF_MARK(7);
    }
    TF_FLUSH();

    if (!first) {      // this workgroup owned at least one tile: its register-resident weight gradients go to the slab
        int tid3 = threadIdx.x;
        asm volatile("" : "+v"(tid3));
        const int wave = tid3 >> 6, lane = tid3 & 63, my_ft = wave % FT, my_sg = wave / FT;
#pragma unroll
        for (int j = 0; j < NW2; ++j)
            *reinterpret_cast<f32x4*>(slabW2 + ((size_t)(my_ft * FT + NW2 * my_sg + j) * 64 + lane) * 4) = aW2[0][j];
#pragma unroll
        for (int q = 0; q < KT1S; ++q) {
            const int j = my_sg + SG * q;
            if (j < kt1) *reinterpret_cast<f32x4*>(slabW1 + ((size_t)(my_ft * kt1 + j) * 64 + lane) * 4) = aW1[q];
        }
#pragma unroll
        for (int k = 0; k < KTBW; ++k) {
            const int ct = wave + NW * k;
            if (ct < ktB) {
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    *reinterpret_cast<f32x4*>(slabB + ((size_t)(i * ktB + ct) * 64 + lane) * 4) = aB[k][i][0];
            }
        }
    }
    // ================= the per-feature scalars (the GEMM gradients are in the slab) =================
    {
        int tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));      // keeps these addresses from being computed (and held) before the tile loop
        const int pf = tid2 % FC;
        atomicAdd(G.w3 + pf, aW3[0]);
        atomicAdd(G.w3 + FC + pf, aW3[1]);
        atomicAdd(G.w3 + 2 * FC + pf, aW3[2]);
        atomicAdd(G.b2 + pf, ab2);
        atomicAdd(G.b1 + pf, ab1);
        if (tid2 < 3) atomicAdd(G.b3 + tid2, ab3);
    }
}

// Sums the workgroups' weight-gradient slabs (fragment order) into the row-major gradient matrices.  Workgroup b
// wrote its slab iff it owned samples, i.e. b * samples_per_wg < total.  One workgroup per 16x16 tile (256 floats
// = 64 float4 columns): thread (group g = tid>>6, lane) adds the slabs b = g, g+4, ... with 16-B loads, the four
// groups meet in LDS.
__global__ __launch_bounds__(256) void wslab_reduce_kernel(const TfShade S, const int* __restrict__ counters,
                                                           int n_wg, const TfShadeGrads G) {
    __shared__ int s_active;
    __shared__ f32x4 part[4][64];
    const int tid = threadIdx.x, grp = tid >> 6, lane = tid & 63;
    if (tid == 0) {
        int run = 0;
        for (int g = 0; g < TF_N_SHARDS; ++g) run += counters[g * TF_SHARD_STRIDE];
        const int q = samples_per_wg(run, n_wg);      // workgroup b of shade_backward owned samples iff b q < total
        s_active = (run + q - 1) / q;
    }
    __syncthreads();
    const int active = s_active;
    const int FT = S.feature_c / 16, kt1 = kpad16(S.in_c) / 16, ktB = kpad16(S.n_app_total) / 16;
    const size_t stride = wslab_floats(S);
    const int tile = blockIdx.x;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    const float* src = G.wslab + (size_t)tile * 256 + lane * 4;
#pragma unroll 4
    for (int b = grp; b < active; b += 4) a += *reinterpret_cast<const f32x4*>(src + (size_t)b * stride);
    part[grp][lane] = a;
    __syncthreads();
    if (grp != 0) return;
    a = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const int c = lane & 15;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int r = 4 * (lane >> 4) + e;
        if (tile < FT * FT) {
            G.w2[(size_t)(16 * (tile / FT) + r) * S.feature_c + 16 * (tile % FT) + c] = a[e];
        } else if (tile < FT * FT + FT * kt1) {
            const int t2 = tile - FT * FT, k = 16 * (t2 % kt1) + c;
            if (k < S.in_c) G.w1[(size_t)(16 * (t2 / kt1) + r) * S.in_c + k] = a[e];
        } else {
            const int t3 = tile - FT * FT - FT * kt1, f = 16 * (t3 / ktB) + r, cc = 16 * (t3 % ktB) + c;
            if (f < S.app_dim && cc < S.n_app_total) G.basis[(size_t)f * S.n_app_total + cc] = a[e];
        }
    }
}

// Direct scatter of dL/dV rows (TensorCP, or VM with binning disabled): one wave per pair of packed samples;
// the gather of the next pair is issued before the atomics of the current one.
__global__ __launch_bounds__(256) void app_direct_scatter_kernel(const TfShade S, const TileSrc src, const TfShadeGrads G) {
    __shared__ int pre[TF_N_SHARDS + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) {
        int run = 0;
        for (int g = 0; g < TF_N_SHARDS; ++g) {
            pre[g] = run;
            run += (src.counters[g * TF_SHARD_STRIDE] + M - 1) / M;
        }
        pre[TF_N_SHARDS] = run;
    }
    __syncthreads();
    const int nit = pair_iters(S.model, S.app);
    const int c0 = S.app.n_comp[0], c1 = S.app.n_comp[1];
    for (int t = blockIdx.x;; t += gridDim.x) {
        int s0, n;
        if (!locate_tile(src, pre, t, s0, n)) break;
        PairVals cur, nxt;
        bool have = false;
        for (int pr = 0; pr < 8; ++pr) {
            const int a = wave * 16 + 2 * pr;
            if (a >= n) break;
            const size_t sA = (size_t)s0 + a, sB = sA + 1;
            const bool hasB = a + 1 < n;
            const float uA[3] = {src.app_xyz[sA * 3], src.app_xyz[sA * 3 + 1], src.app_xyz[sA * 3 + 2]};
            const float uB[3] = {hasB ? src.app_xyz[sB * 3] : uA[0], hasB ? src.app_xyz[sB * 3 + 1] : uA[1],
                                 hasB ? src.app_xyz[sB * 3 + 2] : uA[2]};
            const float* dvA = G.dv_out + sA * S.n_app_total;
            const float* dvB = G.dv_out + (hasB ? sB : sA) * S.n_app_total;
            auto dprod = [&](int s_, int i, int c) {
                const int off_i = S.model == TF_MODEL_VM ? (i == 0 ? 0 : (i == 1 ? c0 : c0 + c1)) : 0;
                return (s_ ? dvB : dvA)[off_i + c];
            };
            for (int it = 0; it < nit; ++it) {
                if (S.model == TF_MODEL_VM)
                    vm_pair_gather(S.app, G.app, S.grid, uA, uB, true, hasB, dprod, lane, it, nxt);
                else
                    cp_pair_gather(S.app, G.app, S.grid, uA, uB, true, hasB, dprod, lane, it, nxt);
                if (have) pair_commit(cur);
                cur = nxt;
                have = true;
            }
        }
        if (have) pair_commit(cur);
    }
}

typedef void (*bwd_fn_t)(const TfShade, const TileSrc, const float*, const TfShadeGrads);

template <int FT, int NB>
bwd_fn_t pick_kt(int kt1, bool rg) {
    if (kt1 <= 4) return rg ? shade_backward_kernel<FT, NB, 4, true> : shade_backward_kernel<FT, NB, 4, false>;
    if (kt1 <= 8) return rg ? shade_backward_kernel<FT, NB, 8, true> : shade_backward_kernel<FT, NB, 8, false>;
    if (kt1 <= 12) return rg ? shade_backward_kernel<FT, NB, 12, true> : shade_backward_kernel<FT, NB, 12, false>;
    return nullptr;
}

bwd_fn_t pick_bwd(const TfShade& S) {
    const int nb = (S.app_dim + 15) / 16, kt1 = kpad16(S.in_c) / 16;
    if (S.head != TF_HEAD_MLP || nb > 2 || kpad16(S.n_app_total) / 16 > 24) return nullptr;
    const bool rg = bwd_lds(S).rg != 0;
    if (S.feature_c == 64) return nb == 1 ? pick_kt<4, 1>(kt1, rg) : pick_kt<4, 2>(kt1, rg);
    if (S.feature_c == 128) return nb == 1 ? pick_kt<8, 1>(kt1, rg) : pick_kt<8, 2>(kt1, rg);
    return nullptr;
}

}  // namespace

#ifdef TF_PHASE_TIMING
static int g_dbg_bwd_wgs = 256;
#endif

extern "C" {

int tf_shade_backward(const TfShade* shade, const float* rays, int ndc, const int* counters, int seg_cap,
                      const int* app_ray, const float* app_xyz, const float* grad_rgb, const TfShadeGrads* grads,
                      tf_stream_t stream) {
    bwd_fn_t fn = pick_bwd(*shade);
    if (!fn) return (int)hipErrorInvalidValue;   // head / width outside the trained configurations
    const BwdLds L = bwd_lds(*shade);
    const size_t bytes = (size_t)L.total * sizeof(float);
    if (bytes > 160 * 1024) return (int)hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    if (!grads->dv_out || !grads->wslab || !shade->w1t || !shade->w2t) return (int)hipErrorInvalidValue;
    TileSrc src{counters, seg_cap, 0, app_ray, app_xyz, rays, ndc};
#ifdef TF_PHASE_TIMING
    const int n_wg = g_dbg_bwd_wgs;
#else
    const int n_wg = 256;
#endif
    hipLaunchKernelGGL(fn, dim3(n_wg), dim3(512), bytes, (hipStream_t)stream, *shade, src, grad_rgb, *grads);
    hipLaunchKernelGGL(wslab_reduce_kernel, dim3((unsigned)(wslab_floats(*shade) / 256)), dim3(256), 0, (hipStream_t)stream, *shade,
                       counters, n_wg, *grads);
    if (grads->direct_scatter)
        hipLaunchKernelGGL(app_direct_scatter_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, *shade, src, *grads);
    return TF_CHECK_LAUNCH();
}

#ifdef TF_PHASE_TIMING
int tf_debug_set_bwd_wgs(int n) { g_dbg_bwd_wgs = n < 1 ? 1 : (n > 256 ? 256 : n); return 0; }
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

/* 1 when tf_shade_backward can differentiate this head: MLP heads of width 64 / 128, app_dim <= 32, an input of at
 * most 192 columns, and a tile (V, X, H1, H2 of 64 samples) that fits the CU's 160 KB of LDS. */
int tf_shade_backward_supported(const TfShade* shade) {
    if (!pick_bwd(*shade)) return 0;
    return (size_t)bwd_lds(*shade).total * sizeof(float) <= 160 * 1024;
}

/* floats the caller must provide in TfShadeGrads.wslab (256 workgroup slabs) */
size_t tf_shade_backward_wslab_floats(const TfShade* shade) { return 256 * wslab_floats(*shade); }

}  // extern "C"
