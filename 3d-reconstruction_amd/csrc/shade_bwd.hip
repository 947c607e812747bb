// shade_bwd.hip — backward of the shading head + appearance lookup (autograd of tensoRF.py:230-263 /
// :388-415 and mlp.py:27-155).   gfx950, wave64, fp32 MFMA.
//
// One persistent 512-thread workgroup (8 waves, 2 per SIMD) per CU walks <= 64-sample chunks of the packed app list.
// The training forward (tf_shade_forward with TfShadeSave) has left, per packed sample, the MLP input row
// X = [feat, view, PE blocks], both hidden layers H1, H2 (after ReLU) and the product row V = plane*line; the sigmoid
// output is the forward's rgb.  NOTHING of the forward is recomputed: the rows are streamed back by LDS-DMA
// (global_load_lds: no registers, asynchronous), one chunk / one phase ahead.  Per chunk:
//     P1  wait for X, H1, H2 (requested during the previous chunk's P7); do = dL/dc . c(1-c) from the forward's colours
//     P3  elementwise on H2: dW3 += do^T H2, dZ2 = (do W3) . [H2>0] (in place of H2), db2 += dZ2
//     P4  dZ1 = (W2^T dZ2) . [H1>0] -> its own buffer, db1 += dZ1;  dW2 += dZ2^T H1          MFMA (two GEMMs, no barrier)
//     P5  dW1 += dZ1^T X;  dX = W1^T dZ1 -> its own buffer (over H1 | H2)                        MFMA (two GEMMs)
//     P6  dfeat = dX[:, :D] + PE'(feat) . dX[:, PE cols];  V (DMA -> LDS, over dZ1)
//     P7  dV = B^T dfeat -> straight from the accumulators to dv_out;  dB += dfeat^T V;  next chunk's requests
// 5 barriers per chunk (17 in round 1).  The weight-gradient GEMMs (k = sample index) read BOTH operands with one wide
// LDS read per lane and 4 samples: lane (r, kq) reads E consecutive columns of row 4t + kq, element e of operand A
// against element e' of operand B feeds accumulator tile (e, e') — rows E_a i + e, columns E_b j + e' of the product —
// so one b128 + one b64 read feed 8 MFMAs (round 1: 36 scalar reads per 32).  The data-gradient GEMMs take their weight
// fragments (L2) one phase ahead, before the barrier in front of them.  The accumulators of dW2, dW1, dB live in
// registers across the chunks of a workgroup and are written once, in fragment order, to the workgroup's slab
// (wslab_reduce_kernel folds the slabs and undoes the fragment order); dW3, db1, db2 are summed in LDS.
#include "tf_shade.h"

using namespace tf;

namespace {

template <int NA, int NBT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NA][NBT]) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NBT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// E (1, 2 or 4) consecutive floats with ONE load
template <int E>
__device__ __forceinline__ void ldv(const float* p, float (&v)[E]) {
    if constexpr (E == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else if constexpr (E == 2) {
        const f32x2 t = *reinterpret_cast<const f32x2*>(p);
        v[0] = t[0]; v[1] = t[1];
    } else {
        v[0] = *p;
    }
}

// acc[ea][eb] += sum_s A[s][a0 + EA i + ea] * B[s][b0 + EB j + eb]  (s = 0 .. 4 ksteps - 1), both operands in LDS,
// sample-major rows: the fp32 "TN" weight-gradient GEMM (dB, P7; the large ones are tn_block_hilo below).  Software-
// pipelined one k-step ahead.
template <int EA, int EB>
__device__ __forceinline__ void tn_block(const float* A, int lda, int a0, const float* B, int ldb, int b0, int ksteps,
                                          f32x4 (&acc)[EA][EB], int lane) {
    const int r = lane & 15, kq = lane >> 4;
    const float* ap = A + kq * lda + a0 + EA * r;
    const float* bp = B + kq * ldb + b0 + EB * r;
    float a[EA], b[EB];
    ldv<EA>(ap, a);
    ldv<EB>(bp, b);
#pragma unroll 2
    for (int t = 0; t < ksteps; ++t) {
        float an[EA], bn[EB];
        const int tn = t + 1 < ksteps ? t + 1 : t;
        ldv<EA>(ap + 4 * tn * lda, an);
        ldv<EB>(bp + 4 * tn * ldb, bn);
#pragma unroll
        for (int ea = 0; ea < EA; ++ea)
#pragma unroll
            for (int eb = 0; eb < EB; ++eb)
                acc[ea][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ea], b[eb], acc[ea][eb], 0, 0, 0);
#pragma unroll
        for (int ea = 0; ea < EA; ++ea) a[ea] = an[ea];
#pragma unroll
        for (int eb = 0; eb < EB; ++eb) b[eb] = bn[eb];
    }
}

// The weight-gradient ("TN") GEMM on the bf16 matrix pipe: acc[ta][tb] += sum_s A[s][a0 + 16 ta + i] * B[s][b0 + 16 tb + j]
// over `ksteps` blocks of 16 samples, A stored as [hi | lo] words (a dZ matrix), B in fp32 (a saved activation matrix,
// split in registers like a weight fragment: tf_shade.h).  The k index of the K = 32 instruction is the SAMPLE: lane
// (idx, g) holds samples 16 t + 4 g .. + 3 of column idx — four 4-byte LDS reads from four consecutive rows per
// fragment.  Rows are 4 (mod 64) words apart (strides 68 / 132 / 164), so the 64 lanes of a read hit the 64 banks once.
// Tiles are plain 16 x 16 blocks (the fp32 tn_block interleaves them): wslab_reduce_kernel maps them back.
template <int TA, int TB>
__device__ __forceinline__ void tn_block_hilo(const float* A, int lda, int a0, const float* B, int ldb, int b0, int ksteps,
                                               f32x4 (&acc)[TA][TB], int lane) {
    const int r = lane & 15, g = lane >> 4;
    const float* ap = A + 4 * g * lda + a0 + r;
    const float* bp = B + 4 * g * ldb + b0 + r;
#pragma unroll 1
    for (int t = 0; t < ksteps; ++t) {
        f32x4 aw[TA], bw[TB];
#pragma unroll
        for (int ta = 0; ta < TA; ++ta)
#pragma unroll
            for (int e = 0; e < 4; ++e) aw[ta][e] = ap[(16 * t + e) * lda + 16 * ta];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int e = 0; e < 4; ++e) bw[tb][e] = bp[(16 * t + e) * ldb + 16 * tb];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            bf16x8 b1, b2;
            split_weight_frag(bw[tb], b1, b2);
#pragma unroll
            for (int ta = 0; ta < TA; ++ta)
                acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aw[ta]), b1, acc[ta][tb], 0, 0, 0);
#pragma unroll
            for (int ta = 0; ta < TA; ++ta)
                acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aw[ta]), b2, acc[ta][tb], 0, 0, 0);
        }
    }
}

struct BwdLds {
    int sv, sx, sh, sf;
    int offV, offH, offX, offF, offDo, offC, offPre, total;
    int wide;    // 1: a V tile does not fit next to the rest -> it is loaded late, over the dZ1 | H space
};
__host__ __device__ inline BwdLds bwd_lds(const TfShade& S) {
    BwdLds L;
    L.sv = kpad16(S.n_app_total) + 4;
    L.sx = kpad16(S.in_c) + 4;
    L.sh = S.feature_c + 4;
    L.sf = 36;                       // dfeat rows (<= 32 used)
    const int hreg = M * (2 * L.sh > L.sx ? 2 * L.sh : L.sx);      // H1 | H2, later dX
    const int consts = 8 * S.feature_c + 8;                         // w3 (3 F); sums of dW3 (3 F), db2, db1; scratch
    for (L.wide = 0; L.wide < 2; ++L.wide) {
        const int vreg = M * (L.wide || L.sh > L.sv ? L.sh : L.sv);   // dZ1, later V (narrow)
        L.offV = 0;
        L.offH = vreg;
        L.offX = L.offH + hreg;
        L.offF = L.offX + M * L.sx;
        L.offDo = L.offF + M * L.sf;
        L.offC = L.offDo + M * 4;
        L.offPre = L.offC + consts;
        L.total = L.offPre + 80;
        if ((size_t)L.total * sizeof(float) <= 160 * 1024) break;
    }
    if (L.wide > 1) L.wide = 1;
    return L;
}
__host__ __device__ inline bool bwd_lds_ok(const TfShade& S, const BwdLds& L) {
    if ((size_t)L.total * sizeof(float) > 160 * 1024) return false;
    return !L.wide || M * L.sv <= L.offF;       // a wide V tile spans the dZ1, H and X regions
}

// ---- weight-gradient tiles (16x16 accumulator fragments) and where their elements belong --------------------
// Per workgroup slab: [dW2 tiles | dW1 tiles | dB tiles], 256 floats each in fragment order [lane][4].
struct WTiles {
    int FC, EA2, EB2, EA1, NTW, NB, ktB;
    int n2, n1, nb;
};
__host__ __device__ inline WTiles wtiles(const TfShade& S, int ntw) {
    WTiles T;
    T.FC = S.feature_c;
    T.EA2 = S.feature_c / 32;      // dW2: A block = FC/2 columns -> EA2 = FC/32 elements per lane (4 | 2)
    T.EB2 = S.feature_c / 64;      //      B block = FC/4 columns (2 | 1)
    T.EA1 = S.feature_c / 64;      // dW1: A block = FC/4 columns (2 | 1)
    T.NTW = ntw;                   //      column tiles per wave (half of the k tiles of layer 1)
    T.NB = (S.app_dim + 15) / 16;
    T.ktB = kpad16(S.n_app_total) / 16;
    T.n2 = 8 * T.EA2 * T.EB2;
    T.n1 = 8 * T.EA1 * T.NTW;
    T.nb = T.NB * T.ktB;
    return T;
}
__host__ __device__ inline int ntw_of(int kt1) {      // compile-time variants: 2, 4, 5, 6 tiles per wave
    const int h = (kt1 + 1) / 2;
    return h <= 2 ? 2 : (h <= 4 ? 4 : h);
}
__host__ __device__ inline size_t wslab_floats(const TfShade& S) {
    const WTiles T = wtiles(S, ntw_of(kpad16(S.in_c) / 16));
    return (size_t)(T.n2 + T.n1 + T.nb) * 256;
}

// The packed sample list is cut EVENLY over the persistent workgroups: workgroup w owns samples [w q, (w+1) q) of the
// shards' concatenation, q = max(64, ceil(S / workgroups)), and walks them in chunks of <= 64.  A chunk may straddle
// a shard boundary, so it has two pieces.
struct Chunk {
    int s0, n0, s1, n1;      // packed positions / lengths of the two pieces (n1 may be 0)
    __device__ __forceinline__ int n() const { return n0 + n1; }
    __device__ __forceinline__ size_t at(int k) const { return k < n0 ? (size_t)s0 + k : (size_t)s1 + (k - n0); }
};
__host__ __device__ inline int samples_per_wg(int total, int n_wg) {
    const int q = (total + n_wg - 1) / n_wg;
    return q < M ? M : q;
}
// spre: exclusive prefix of the shards' sample counts (TF_N_SHARDS + 1 entries); v in [0, v_end).  Called by whole waves:
// lane k looks at shard k and a ballot finds the last shard starting at or before v (one LDS read per lane; a 63-step
// unrolled scan keeps 63 registers live).
__device__ __forceinline__ bool locate_chunk(const TileSrc& src, const int* spre, int v, int v_end, Chunk& c) {
    c.s0 = c.n0 = c.s1 = c.n1 = 0;
    if (v >= v_end) return false;
    static_assert(TF_N_SHARDS == 64, "one shard per lane");
    const int g = __builtin_popcountll(__ballot(spre[threadIdx.x & 63] <= v)) - 1;      // spre[0] = 0 <= v
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

// Global -> LDS without registers (global_load_lds_dwordx4): the 64 x (4 * sq)-float row image at `dst` is filled in
// 1-KiB pieces (one wave instruction: lane l lands at piece base + 16 l bytes); row r < n comes from src + at(r) * w
// floats, w = 4 wq <= 4 sq data floats per row.  Pad quads and rows >= n are left alone (EXEC-masked lanes do not
// write): callers keep stale-but-finite data there.  Asynchronous: the issuing wave's vmcnt covers it.
template <typename AtFn>
__device__ __forceinline__ void dma_rows(const float* __restrict__ src, int wq, float* dst, int sq, float inv_sq, int n,
                                         int wave, int lane, AtFn at) {
    const int pieces = sq;             // 64 rows * sq quads / 64 lanes
    for (int p = wave; p < pieces; p += 8) {
        const int q = p * 64 + lane;
        int row, c4;
        row_quad(q, sq, inv_sq, row, c4);
        if (row < n && c4 < wq)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + at(row) * (size_t)(4 * wq) + 4 * c4),
                                             (__attribute__((address_space(3))) void*)(dst + p * 256), 16, 0, 0);
    }
}

// FT = feature_c/16 hidden feature tiles (4 or 8), NB = ceil(app_dim/16) (1..2), NTW = layer-1 k tiles per wave half
// (2, 4, 5, 6 <-> up to 4, 8, 10, 12 k tiles).  WIDE: see BwdLds.  512 threads = 8 waves = 2 per SIMD.
// Wave w: hidden feature tile ft = w % FT, sample group sg = w / FT (SG = 8/FT groups of NSW = 4/SG sample tiles).
// MODE 0: V tile beside the rest, dV's basis operand staged in LDS (X region) | 1: WIDE | 2: as 0, basis from L2 (it does
// not fit the X region: narrow MLP inputs)
template <int FT, int NB, int NTW, int MODE>
__global__ __launch_bounds__(512, 2) void shade_backward_kernel(const TfShade S, const TileSrc src,
                                                                 const float* __restrict__ grad_rgb,
                                                                 const TfShadeGrads G) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr bool WIDE = MODE == 1, BLDS = MODE == 0;
    constexpr int NT = 512, NW = 8, SG = NW / FT, NSW = 4 / SG;
    constexpr int FCc = 16 * FT, EA2 = FCc / 32, EB2 = FCc / 64, EA1 = FCc / 64;
    constexpr int KTBW = 3;                // dB: column tiles wave, wave + 8, wave + 16 (n_app_total <= 384)
    const BwdLds L = bwd_lds(S);
    float* DZ1 = lds + L.offV;            // dZ1 [64][sh]; later V [64][sv]
    float* V = lds + L.offV;
    float* H1 = lds + L.offH;             // [64][sh]
    float* H2 = H1 + M * L.sh;            // [64][sh]; dZ2 in place
    float* DX = lds + L.offH;             // dX [64][sx] over H1 | H2
    float* X = lds + L.offX;              // [64][sx]
    float* Fd = lds + L.offF;             // dfeat [64][sf]
    float* dO = lds + L.offDo;            // [64][4]
    float* cw3 = lds + L.offC;            // [3][FC]
    float* sW3 = cw3 + 3 * FCc;           // [3][FC] dW3, [FC] db2, [FC] db1: this workgroup's sums over its chunks
    float* sb2 = sW3 + 3 * FCc;
    float* sb1 = sb2 + FCc;
    float* cred = sb1 + FCc;              // [8] reduction scratch
    int* pre = reinterpret_cast<int*>(lds + L.offPre);
    const int tid0 = threadIdx.x;
    const int kp1 = kpad16(S.in_c), kt1 = kp1 / 16, kpB = kpad16(S.n_app_total), ktB = kpB / 16;
    const int xq4 = kp1 / 4, nat = S.n_app_total, vq4 = nat >> 2;
    const bool v_vec = (nat & 3) == 0;            // V / dV rows are 16-B aligned
    const float* __restrict__ xs = G.x_saved;
    float* __restrict__ dv = G.dv_out;            // V rows on entry, dL/dV rows on exit

    // ---- slab of this workgroup and the register-resident accumulators
    const WTiles T = wtiles(S, NTW);
    float* slab = G.wslab + (size_t)blockIdx.x * ((size_t)(T.n2 + T.n1 + T.nb) * 256);
    bool first = true;
    f32x4 aW2[EA2][EB2];
    zero_acc(aW2);
    f32x4 aW1[EA1][NTW];
    zero_acc(aW1);
    f32x4 aB[KTBW][NB];
#pragma unroll
    for (int k = 0; k < KTBW; ++k)
#pragma unroll
        for (int e = 0; e < NB; ++e) aB[k][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float ab3 = 0.f;
    float tW3[3] = {0.f, 0.f, 0.f}, tb2 = 0.f;      // dW3 / db2 of thread (feature tid % FC, sample slice tid / FC)
    float ab1[4] = {0.f, 0.f, 0.f, 0.f};             // db1 of this lane's 4 features (accumulator layout), all chunks

    shard_prefix(src.counters, src.seg_cap, pre, tid0);
    for (int i = tid0; i < 3 * FCc; i += NT) cw3[i] = S.w3[i];
    for (int i = tid0; i < 5 * FCc; i += NT) sW3[i] = 0.f;
    __syncthreads();
    const int q_wg = samples_per_wg(pre[TF_N_SHARDS], (int)gridDim.x);
    const int v_end = min(pre[TF_N_SHARDS], ((int)blockIdx.x + 1) * q_wg);

    // ---- operands of the NEXT chunk, requested while the current one is processed: the X rows travel global -> LDS
    // by DMA (no registers); thread (sample tid>>2, channel tid&3) holds dL/dc and the forward's colour
    const int sxq = L.sx >> 2, svq = L.sv >> 2, shq = L.sh >> 2;
    const float inv_sxq = 1.f / (float)sxq, inv_svq = 1.f / (float)svq, inv_shq = 1.f / (float)shq;
    float n_g = 0.f, n_c = 0.f;
    auto fetch_x = [&](const Chunk& ck, int tid, bool dma) {
        const int n = ck.n();
        n_g = n_c = 0.f;        // (requested ahead of the DMA pieces: P1 needs them first)
        if (tid < 4 * M && (tid & 3) < 3 && (tid >> 2) < n) {
            const size_t p = ck.at(tid >> 2) * 3 + (tid & 3);
            n_g = grad_rgb[p];
            n_c = G.rgb_fwd[p];
        }
        // The CU takes in ~17 B per cycle: a chunk's 105 KB of rows requested in one burst stall the requesting waves for
        // ~6 k cycles (the queue backs up into the issue).  So each tile is requested just ahead of the phase before its
        // first use, beside that phase's arithmetic: H2 here (P3 needs it), H1 at the start of P3 (P4), X in P4 (P5).
        if (dma) {
            const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
            dma_rows(G.h2_saved, FCc / 4, H2, shq, inv_shq, n, w, l, [&](int r) { return ck.at(r); });
        }
    };
    // rows past a chunk's end keep what the previous chunk left there: finite, and multiplied by dZ = 0.  Before the
    // first chunk the X region (and, for the scalar V path, nothing else) must not hold NaN bit patterns
    for (int i = tid0; i < M * L.sx; i += NT) X[i] = 0.f;
    for (int i = tid0; i < L.offX - L.offH; i += NT) H1[i] = 0.f;
    if (!WIDE || true) {
        for (int i = tid0; i < (WIDE ? L.offF : M * L.sv); i += NT) V[i] = 0.f;      // V / dZ1 region likewise
    }
    __syncthreads();
    {
        Chunk c1;
        if (locate_chunk(src, pre, (int)blockIdx.x * q_wg, v_end, c1)) fetch_x(c1, tid0, true);
    }

    TF_T0();
    for (int v = (int)blockIdx.x * q_wg;;) {
        Chunk ck;
        if (!locate_chunk(src, pre, v, v_end, ck)) break;
        const int n = ck.n();
        v += n;
        // Thread coordinates are re-derived per chunk from an opaque copy of the thread id: otherwise the compiler
        // hoists every phase's per-thread addresses out of the chunk loop and runs out of registers.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        // the wave index through readfirstlane: wave-dependent branches (which tiles a wave owns) are then scalar branches
        // instead of per-lane selects over whole accumulator tiles
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const int my_ft = wave % FT, my_sg = wave / FT, s_base = my_sg * NSW * 16;
        const int lc = lane & 15, lg = lane >> 4;

        // ================= P1: do into LDS; the rows requested during the previous chunk's P7 have landed =========
        if (tid < 4 * M) {
            const float d = n_g * (n_c * (1.f - n_c));       // dL/dc . sigmoid'   (mlp.py:67)
            dO[tid] = (tid & 3) < 3 ? d : 0.f;
            ab3 += d;
        }
        f32x4 fr3[FT][1];                                    // weight fragments of P4, in flight across two barriers
        load_a_frags<1, FT>(S.w2t, FCc, 16 * my_ft, FT, lane, fr3);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(FT) : "memory");     // this wave's DMA pieces of H2 are in LDS
        lds_barrier();
        TF_MARK(0);

        // ================= P3: dW3 += do^T H2, dZ2 = (do W3) . [H2 > 0] in place of H2, db2 += dZ2 =================
        dma_rows(G.h1_saved, FCc / 4, H1, shq, inv_shq, n, wave, lane, [&](int r) { return ck.at(r); });     // for P4
        {   // thread -> (feature pf, slice of 64 / (512 / FC) samples); LDS sums (the slices of a feature meet there)
            const int pf = tid % FCc, slice = tid / FCc;
            constexpr int SPAN = M / (NT / FCc);
            const float w0 = cw3[pf], w1 = cw3[FCc + pf], w2 = cw3[2 * FCc + pf];
            constexpr int BT = SPAN < 8 ? SPAN : 8;      // samples per batch: all reads of a batch in flight together
#pragma unroll 1
            for (int s0 = 0; s0 < SPAN; s0 += BT) {
                f32x4 d[BT];
                float h[BT];
                float* hp = H2 + (slice * SPAN + s0) * L.sh + pf;
#pragma unroll
                for (int i = 0; i < BT; ++i) {
                    d[i] = *reinterpret_cast<const f32x4*>(dO + 4 * (slice * SPAN + s0 + i));
                    h[i] = hp[i * L.sh];
                }
#pragma unroll
                for (int i = 0; i < BT; ++i) {
                    tW3[0] = fmaf(d[i][0], h[i], tW3[0]);
                    tW3[1] = fmaf(d[i][1], h[i], tW3[1]);
                    tW3[2] = fmaf(d[i][2], h[i], tW3[2]);
                    const float dz = h[i] > 0.f ? fmaf(d[i][2], w2, fmaf(d[i][1], w1, d[i][0] * w0)) : 0.f;
                    hp[i * L.sh] = pack_hilo(dz);          // dZ2 as [hi | lo] words: operand of the bf16 data-gradient product
                    tb2 += dz;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces of H1 are in LDS
        lds_barrier();
        TF_MARK(2);

        // ================= P4: dZ1 = (W2^T dZ2) . [H1 > 0] -> DZ1, db1;  dW2 += dZ2^T H1 =================
        f32x4 frx[1][FT][1];       // dX (P5): weight fragments of the wave's current work item
        {
            f32x4 acc[1][NSW];
            zero_acc(acc);
            mma_frags_hilo<NSW, FT>(fr3, H2, L.sh, s_base, FT, acc, lane);
            if (wave < 2 * kt1) load_a_frags<1, FT>(S.w1t, FCc, 16 * (wave >> 1), FT, lane, frx[0]);
            const int f = 16 * my_ft + 4 * lg;
#pragma unroll
            for (int j = 0; j < NSW; ++j) {
                const int s = s_base + 16 * j + lc;
                const f32x4 h = *reinterpret_cast<const f32x4*>(H1 + s * L.sh + f);
                f32x4 dz;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dz[e] = h[e] > 0.f ? acc[0][j][e] : 0.f;
                    ab1[e] += dz[e];
                    dz[e] = pack_hilo(dz[e]);              // dZ1 leaves as [hi | lo] words too (dX = W1^T dZ1 below)
                }
                *reinterpret_cast<f32x4*>(DZ1 + s * L.sh + f) = dz;
            }

            TF_MARK(3);
            dma_rows(xs, xq4, X, sxq, inv_sxq, n, wave, lane, [&](int r) { return ck.at(r); });     // for P5, lands behind dW2
            // dW2[f2][f1]: wave -> (f2 block of FC/2, f1 block of FC/4)
            tn_block_hilo<EA2, EB2>(H2, L.sh, (FCc / 2) * (wave >> 2), H1, L.sh, (FCc / 4) * (wave & 3), M / 16, aW2, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces of X are in LDS
        lds_barrier();
        TF_MARK(4);

        // ================= P5: dW1 += dZ1^T X;  dX = W1^T dZ1 -> DX (over H1 | H2) =================
        {
            // dW1[f][k]: wave -> (f block of FC/4 columns of dZ1, half of the k tiles of X)
            tn_block_hilo<EA1, NTW>(DZ1, L.sh, (FCc / 4) * (wave >> 1), X, L.sx, 16 * NTW * (wave & 1), M / 16, aW1, lane);
        }
        TF_MARK(5);
        // dX[k][s] = sum_f W1[f][k] dZ1[s][f].  Work items are (k tile, pair of sample tiles): 2 kt1 items dealt
        // round-robin (waves w and w + 4 share a SIMD, so 2.5 items per wave are 5 per SIMD)
#pragma unroll
        for (int round = 0; round < 3; ++round) {       // 2 kt1 <= 24 items over 8 waves
            const int it = wave + NW * round;
            if (it < 2 * kt1) {
                const int kt = it >> 1, sp = it & 1;
                if (round > 0) load_a_frags<1, FT>(S.w1t, FCc, 16 * kt, FT, lane, frx[0]);
                f32x4 acc[1][2];
                zero_acc(acc);
                mma_frags_hilo<2, FT>(frx[0], DZ1, L.sh, 32 * sp, FT, acc, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<f32x4*>(DX + (32 * sp + 16 * j + lc) * L.sx + 16 * kt + 4 * lg) = acc[0][j];
            }
        }
        lds_barrier();
        TF_MARK(6);

        // ================= P6: dfeat -> Fd;  V -> LDS (over dZ1) =================
        if (!WIDE && v_vec)      // V rows of this chunk: global -> LDS by DMA, landing behind the dfeat arithmetic
            dma_rows(dv, vq4, V, svq, inv_svq, n, wave, lane, [&](int r) { return ck.at(r); });
        {   // dfeat = dX[:, :D] + PE'(feat): d/dx [sin(x 2^k) m_s] = cos(.) 2^k m_s,  d/dx [cos(.) m_c] = -sin(.) 2^k m_c
            // Thread -> column d = tid % (16 NB) of the samples tid / (16 NB) + SPP i: the SI samples of a thread advance
            // together through the encoding terms, so SI independent LDS reads are in flight per term (sample by sample the
            // phase was a chain of LDS latencies: runtime loop bounds, one accumulator).  Same terms, same order per sample.
            constexpr int DC = 16 * NB, SPP = NT / DC, SI = M / SPP;
            const int d = tid % DC, s0 = tid / DC;
            float gsum[SI];
#pragma unroll
            for (int i = 0; i < SI; ++i) gsum[i] = 0.f;
            if (d < S.app_dim) {
#pragma unroll
                for (int i = 0; i < SI; ++i) gsum[i] = DX[(s0 + SPP * i) * L.sx + d];
                int off = S.app_dim + 3;
                for (int b = 0; b < S.n_pe; ++b) {
                    const int F = S.pe[b].freqs;
                    const int D = S.pe[b].src == TF_SRC_FEAT ? S.app_dim : 3;
                    if (S.pe[b].src == TF_SRC_FEAT) {
                        const float* mk = S.pe[b].mask;
                        if (!mk) {     // unmasked: X's PE columns ARE sin / cos of these arguments (saved by the forward)
                            float fr = 1.f;
                            for (int k = 0; k < F; ++k) {
                                const int ci = d * F + k;
                                float a0[SI], a1[SI], c0[SI], c1[SI];
#pragma unroll
                                for (int i = 0; i < SI; ++i) {
                                    const int row = (s0 + SPP * i) * L.sx;
                                    a0[i] = DX[row + off + ci];
                                    c0[i] = X[row + off + D * F + ci];
                                    a1[i] = DX[row + off + D * F + ci];
                                    c1[i] = X[row + off + ci];
                                }
#pragma unroll
                                for (int i = 0; i < SI; ++i) {
                                    gsum[i] += a0[i] * (c0[i] * fr);
                                    gsum[i] -= a1[i] * (c1[i] * fr);
                                }
                                fr *= 2.f;
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < SI; ++i) {
                                const int row = (s0 + SPP * i) * L.sx;
                                const float fv = X[row + d];
                                const bool big = !(ldexpf(fabsf(fv), F - 1) < 8192.f);
                                float fr = 1.f;
                                for (int k = 0; k < F; ++k) {
                                    float sn, cs;
                                    if (__builtin_expect(big, 0)) pe_sincos(fv * fr, &sn, &cs);
                                    else pe_sincos_fast(fv * fr, &sn, &cs);
                                    const int ci = d * F + k;
                                    gsum[i] += DX[row + off + ci] * (cs * fr * mk[ci]);
                                    gsum[i] -= DX[row + off + D * F + ci] * (sn * fr * mk[D * F + ci]);
                                    fr *= 2.f;
                                }
                            }
                        }
                    }
                    off += 2 * D * F;
                }
            }
#pragma unroll
            for (int i = 0; i < SI; ++i) Fd[(s0 + SPP * i) * L.sf + d] = gsum[i];
        }
        if (!WIDE && !v_vec) {
            for (int smp = wave; smp < M; smp += NW)
                for (int c = lane; c < nat; c += 64) V[smp * L.sv + c] = smp < n ? dv[ck.at(smp) * nat + c] : 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces of V are in LDS
        // basis fragments of the wave's first dV unit (P7), in flight across the barrier
        const int n_units = 4 * ((ktB + 3) >> 2);
        // dV's A operand, the packed basis [16 NB][kpB] (+ 64 floats of tail), in LDS for P7 when it fits the X region (free
        // once P6 has read X): every workgroup used to re-read its fragments from L2 for every chunk — 98 KB per chunk at
        // config 2 on a CU that takes in ~17 B per cycle, and the second dV unit of waves 4..7 waited for them
        const int basis_fl = 16 * NB * kpB + 64;
        constexpr bool b_lds = BLDS;          // (the launch picks MODE 0 only when basis_fl <= M * sx: basis_in_lds())
        f32x4 bfr[4 * NB];
        if constexpr (!b_lds) {
            const int u0 = NW - 1 - wave, r = lane & 15, kq = lane >> 4;
            const float* np = S.basis + (size_t)kq * kpB + 64 * (u0 >> 2) + 4 * r;
#pragma unroll
            for (int t = 0; t < 4 * NB; ++t)
                bfr[t] = u0 < n_units ? *reinterpret_cast<const f32x4*>(np + (size_t)4 * t * kpB) : (f32x4){0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
        if (WIDE) {     // the V tile needs the dZ1 | H space, which dfeat has just finished reading
            for (int smp = wave; smp < M; smp += NW)
                for (int c = lane; c < nat; c += 64) V[smp * L.sv + c] = smp < n ? dv[ck.at(smp) * nat + c] : 0.f;
            lds_barrier();
        }
        TF_MARK(7);
        if constexpr (b_lds) {
            // ---- P7, basis in LDS: [basis DMA | dB tile 0] -> barrier -> next-chunk requests -> dV from LDS -> other dB tiles
            float* BL = X;
            for (int p = wave; 256 * p < basis_fl; p += NW)
                if (256 * p + 4 * lane < basis_fl)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(S.basis + 256 * p + 4 * lane),
                                                     (__attribute__((address_space(3))) void*)(BL + 256 * p), 16, 0, 0);
            {
                f32x4 (&tile)[NB][1] = reinterpret_cast<f32x4 (&)[NB][1]>(aB[0]);
                if (wave < ktB) tn_block<NB, 1>(Fd, L.sf, 0, V, L.sv, 16 * wave, M / 4, tile, lane);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the basis are in LDS
            lds_barrier();
            {
                Chunk c1;
                if (locate_chunk(src, pre, v, v_end, c1)) fetch_x(c1, tid, true);
                else n_g = n_c = 0.f;
            }
            TF_MARK(10);
            {
                const int r = lane & 15, kq = lane >> 4;
                for (int u = NW - 1 - wave; u < n_units; u += NW) {
                    const int st = u & 3, cb = 64 * (u >> 2);
                    const float* bp = Fd + (16 * st + r) * L.sf + kq;
                    const float* ap = BL + kq * kpB + cb + 4 * r;
                    const int s = 16 * st + lc;
                    f32x4 acc[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int h = 0; h < 4 * NB; h += 4) {      // four k-steps of operands at a time (16 + 4 registers)
                        float bq[4];
                        f32x4 af[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            bq[t] = bp[4 * (h + t)];
                            af[t] = *reinterpret_cast<const f32x4*>(ap + 4 * (h + t) * kpB);
                        }
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][e], bq[t], acc[e], 0, 0, 0);
                    }
                    // The accumulators leave through a [16][64] tile of this wave in LDS (H1's space, or the X region
                    // behind the basis): a lane holds 16 B pieces 64 B apart in FOUR rows' worth of lines per store
                    // instruction — 64 partial lines — and the stores backed up into the issue (3.5 k cycles per unit);
                    // read back row-wise, 16 lanes write 256 contiguous bytes.
                    float* stg = wave < 5 ? H1 + wave * (16 * 68) : X + ((basis_fl + 3) & ~3) + (wave - 5) * (16 * 68);
                    (void)s;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        *reinterpret_cast<f32x4*>(stg + lc * 68 + 16 * lg + 4 * reg) =
                            (f32x4){acc[0][reg], acc[1][reg], acc[2][reg], acc[3][reg]};
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        const int row = 4 * i4 + (lane >> 4), c = cb + 4 * (lane & 15), sr = 16 * st + row;
                        const f32x4 val = *reinterpret_cast<const f32x4*>(stg + row * 68 + 4 * (lane & 15));
                        if (sr < n) {
                            float* o = dv + ck.at(sr) * nat;
                            if (v_vec && c + 3 < nat) {
                                *reinterpret_cast<f32x4*>(o + c) = val;
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (c + e < nat) o[c + e] = val[e];
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            TF_MARK(8);
#pragma unroll
            for (int k = 1; k < KTBW; ++k) {
                f32x4 (&tile)[NB][1] = reinterpret_cast<f32x4 (&)[NB][1]>(aB[k]);
                if (wave + NW * k < ktB) tn_block<NB, 1>(Fd, L.sf, 0, V, L.sv, 16 * (wave + NW * k), M / 4, tile, lane);
            }
        } else {
        // operands of the next chunk: its X rows go straight into the X region, which nothing reads any more (a WIDE V
        // tile lies over it: there the request waits until P7 is done)
        {
            Chunk c1;
            if (locate_chunk(src, pre, v, v_end, c1)) fetch_x(c1, tid, !WIDE);
            else n_g = n_c = 0.f;
        }

        TF_MARK(10);
        // ================= P7: dV = B^T dfeat -> dv_out;  dB += dfeat^T V =================
        {   // dV[s][c] = sum_f B[f][c] dfeat[s][f]: units (group of 4 column tiles, sample tile), dealt from the last
            // wave down (the low waves own the extra dB tiles).  A = packed basis rows f (k index), read 4 columns per
            // lane (the first unit's fragments `bfr` were requested before P6's barrier; the next unit's replace them
            // behind the MFMAs); B = Fd[s][f].  The last group may reach past the packed row (TfShade.basis carries 64
            // floats of tail padding): those columns are computed and dropped.  The accumulators go straight to dv_out,
            // ahead of the dB loop, so that the stores have retired when the next chunk waits for its X rows.
            const int r = lane & 15, kq = lane >> 4;
            for (int u = NW - 1 - wave; u < n_units; u += NW) {
                const int st = u & 3, cb = 64 * (u >> 2), un = u + NW;
                const float* bp = Fd + (16 * st + r) * L.sf + kq;
                const int s = 16 * st + lc;
                float bq[4 * NB];
#pragma unroll
                for (int t = 0; t < 4 * NB; ++t) bq[t] = bp[4 * t];
                f32x4 acc[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (un < n_units) {     // the next unit's fragments replace this unit's, k-step by k-step
                    const float* np = S.basis + (size_t)kq * kpB + 64 * (un >> 2) + 4 * r;
#pragma unroll
                    for (int t = 0; t < 4 * NB; ++t) {
                        const f32x4 a = bfr[t];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], bq[t], acc[e], 0, 0, 0);
                        bfr[t] = *reinterpret_cast<const f32x4*>(np + (size_t)4 * t * kpB);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4 * NB; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[t][e], bq[t], acc[e], 0, 0, 0);
                }
                if (s < n) {     // element (e, reg): column cb + 4 (4 lg + reg) + e
                    float* o = dv + ck.at(s) * nat;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int c = cb + 16 * lg + 4 * reg;
                        if (v_vec && c + 3 < nat) {
                            *reinterpret_cast<f32x4*>(o + c) = (f32x4){acc[0][reg], acc[1][reg], acc[2][reg], acc[3][reg]};
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (c + e < nat) o[c + e] = acc[e][reg];
                        }
                    }
                }
            }
        }
        TF_MARK(8);
        // dB[f][c] += sum_s dfeat[s][f] V[s][c]: A = Fd (NB elements per lane: f = NB i + e), B = V column tiles wave,
        // wave + 8, wave + 16 — one pipelined loop per tile the wave owns
#pragma unroll
        for (int k = 0; k < KTBW; ++k) {
            f32x4 (&tile)[NB][1] = reinterpret_cast<f32x4 (&)[NB][1]>(aB[k]);
            if (wave + NW * k < ktB) tn_block<NB, 1>(Fd, L.sf, 0, V, L.sv, 16 * (wave + NW * k), M / 4, tile, lane);
        }
        }
        first = false;
        TF_MARK(9);
        // no barrier here: the next chunk's P1 writes X and dO only, which nothing in P7 reads (a WIDE V tile lies over X)
        if (WIDE) {
            lds_barrier();
            Chunk c1;
            if (locate_chunk(src, pre, v, v_end, c1))
                dma_rows(G.h2_saved, FCc / 4, H2, shq, inv_shq, c1.n(), wave, lane, [&](int r) { return c1.at(r); });
        }
    }
    TF_FLUSH();

    int tid3 = threadIdx.x;
    asm volatile("" : "+v"(tid3));
    if (!first) {      // this workgroup owned at least one chunk: its register-resident weight gradients go to the slab
        const int wave = __builtin_amdgcn_readfirstlane(tid3 >> 6), lane = tid3 & 63;
        float* sl = slab;
#pragma unroll
        for (int ea = 0; ea < EA2; ++ea)
#pragma unroll
            for (int eb = 0; eb < EB2; ++eb)
                *reinterpret_cast<f32x4*>(sl + ((size_t)((wave * EA2 + ea) * EB2 + eb) * 64 + lane) * 4) = aW2[ea][eb];
        sl += (size_t)T.n2 * 256;
#pragma unroll
        for (int ea = 0; ea < EA1; ++ea)
#pragma unroll
            for (int t = 0; t < NTW; ++t)
                *reinterpret_cast<f32x4*>(sl + ((size_t)((wave * EA1 + ea) * NTW + t) * 64 + lane) * 4) = aW1[ea][t];
        sl += (size_t)T.n1 * 256;
#pragma unroll
        for (int k = 0; k < KTBW; ++k) {
            const int ct = wave + NW * k;
            if (ct < ktB) {
#pragma unroll
                for (int e = 0; e < NB; ++e)
                    *reinterpret_cast<f32x4*>(sl + ((size_t)(e * ktB + ct) * 64 + lane) * 4) = aB[k][e];
            }
        }
    }
    // ================= the per-feature vectors (summed in LDS over this workgroup's chunks): one atomic per entry ======
    {
        {   // the threads' / lanes' sums meet in LDS (once per workgroup)
            const int pf = tid3 % FCc, lane = tid3 & 63, lc = lane & 15;
            const int f = 16 * ((tid3 >> 6) % FT) + 4 * (lane >> 4);
            atomicAdd(sW3 + pf, tW3[0]);
            atomicAdd(sW3 + FCc + pf, tW3[1]);
            atomicAdd(sW3 + 2 * FCc + pf, tW3[2]);
            atomicAdd(sb2 + pf, tb2);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v4 = row16_sum(ab1[e]);
                if (lc == 0) atomicAdd(sb1 + f + e, v4);
            }
        }
        __syncthreads();
        for (int i = tid3; i < 3 * FCc; i += NT) atomicAdd(G.w3 + i, sW3[i]);
        if (tid3 < FCc) {
            atomicAdd(G.b2 + tid3, sb2[tid3]);
            atomicAdd(G.b1 + tid3, sb1[tid3]);
        }
        // db3: thread (sample, channel) sums over its chunks -> LDS -> one atomic per channel
        __syncthreads();
        if (tid3 < 4) cred[tid3] = 0.f;
        __syncthreads();
        if (tid3 < 4 * M && (tid3 & 3) < 3) atomicAdd(cred + (tid3 & 3), ab3);
        __syncthreads();
        if (tid3 < 3) atomicAdd(G.b3 + tid3, cred[tid3]);
    }
}

// Sums the workgroups' weight-gradient slabs (fragment order) into the row-major gradient matrices.  Workgroup b
// wrote its slab iff it owned samples, i.e. b * samples_per_wg < total.  One workgroup per 16x16 fragment tile (256
// floats = 64 float4): thread (group g = tid>>6, lane) adds the slabs b = g, g+16, ... with 16-B loads, the sixteen
// groups meet in LDS; the element's (row, column) is recovered from the tile's place in the slab (see WTiles).
__global__ __launch_bounds__(1024) void wslab_reduce_kernel(const TfShade S, const int* __restrict__ counters, int seg_cap,
                                                            int n_wg, int ntw, const TfShadeGrads G) {
    constexpr int NG = 16;      // thread groups per tile: 256 slabs are 16 loads per thread, all in flight at once (with four
                                // groups a thread walked 64 slabs four at a time: 15 us of load latency for 41 MB)
    __shared__ int s_active;
    __shared__ f32x4 part[NG][64];
    const int tid = threadIdx.x, grp = tid >> 6, lane = tid & 63;
    if (tid < 64) {
        static_assert(TF_N_SHARDS == 64, "one shard per lane");
        int run = min(counters[tid * TF_SHARD_STRIDE], seg_cap);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) run += __shfl_xor(run, o, 64);
        if (tid == 0) {
            const int q = samples_per_wg(run, n_wg);      // workgroup b of shade_backward owned samples iff b q < total
            s_active = (run + q - 1) / q;
        }
    }
    __syncthreads();
    const int active = s_active;
    const WTiles T = wtiles(S, ntw);
    const size_t stride = (size_t)(T.n2 + T.n1 + T.nb) * 256;
    // (two workgroups per tile, each summing every other slab and adding its half with atomics, so that no CU idles:
    // 11.8 us against 12.6 — the kernel reads 41 MB the shading backward has just written, ~3.3 TB/s either way)
    const int tile = blockIdx.x;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    const float* src = G.wslab + (size_t)tile * 256 + lane * 4;
#pragma unroll 16
    for (int b = grp; b < active; b += NG) a += *reinterpret_cast<const f32x4*>(src + (size_t)b * stride);
    part[grp][lane] = a;
    __syncthreads();
    if (grp != 0) return;
    a = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
#pragma unroll
    for (int g = 4; g < NG; g += 4) a += (part[g][lane] + part[g + 1][lane]) + (part[g + 2][lane] + part[g + 3][lane]);
    const int j = lane & 15;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = 4 * (lane >> 4) + e;
        if (tile < T.n2) {            // dW2 tile (wave, ea, eb)
            const int eb = tile % T.EB2, ea = (tile / T.EB2) % T.EA2, w = tile / (T.EA2 * T.EB2);
            const int f2 = (T.FC / 2) * (w >> 2) + 16 * ea + i, f1 = (T.FC / 4) * (w & 3) + 16 * eb + j;
            G.w2[(size_t)f2 * T.FC + f1] = a[e];
        } else if (tile < T.n2 + T.n1) {   // dW1 tile (wave, ea, t)
            const int t2 = tile - T.n2, t = t2 % T.NTW, ea = (t2 / T.NTW) % T.EA1, w = t2 / (T.NTW * T.EA1);
            const int f = (T.FC / 4) * (w >> 1) + 16 * ea + i, k = 16 * T.NTW * (w & 1) + 16 * t + j;
            if (k < S.in_c) G.w1[(size_t)f * S.in_c + k] = a[e];
        } else {                      // dB tile (e_f, column tile)
            const int t3 = tile - T.n2 - T.n1, ct = t3 % T.ktB, ef = t3 / T.ktB;
            const int f = T.NB * i + ef, cc = 16 * ct + j;
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
            run += (min(src.counters[g * TF_SHARD_STRIDE], src.seg_cap) + M - 1) / M;
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
bwd_fn_t pick_ntw(int ntw, int mode) {
    switch (ntw) {
        case 2: return mode == 1 ? shade_backward_kernel<FT, NB, 2, 1> : (mode == 0 ? shade_backward_kernel<FT, NB, 2, 0> : shade_backward_kernel<FT, NB, 2, 2>);
        case 4: return mode == 1 ? shade_backward_kernel<FT, NB, 4, 1> : (mode == 0 ? shade_backward_kernel<FT, NB, 4, 0> : shade_backward_kernel<FT, NB, 4, 2>);
        case 5: return mode == 1 ? shade_backward_kernel<FT, NB, 5, 1> : (mode == 0 ? shade_backward_kernel<FT, NB, 5, 0> : shade_backward_kernel<FT, NB, 5, 2>);
        case 6: return mode == 1 ? shade_backward_kernel<FT, NB, 6, 1> : (mode == 0 ? shade_backward_kernel<FT, NB, 6, 0> : shade_backward_kernel<FT, NB, 6, 2>);
    }
    return nullptr;
}

bwd_fn_t pick_bwd(const TfShade& S) {
    const int nb = (S.app_dim + 15) / 16, kt1 = kpad16(S.in_c) / 16;
    if (S.head != TF_HEAD_MLP || nb > 2 || kt1 > 12 || kpad16(S.n_app_total) / 16 > 24) return nullptr;
    const BwdLds L = bwd_lds(S);
    if (!bwd_lds_ok(S, L)) return nullptr;
    const int ntw = ntw_of(kt1);
    // the basis image of P7 (16 nb rows of kpad16(n_app_total) floats + 64 of tail) in the X region?
    // ... and the dV staging tiles (16 x 68 floats per wave: five in H1's space, three behind the basis)?
    const int basis_fl = 16 * nb * kpad16(S.n_app_total) + 64;
    const bool basis_fits = ((basis_fl + 3) & ~3) + 3 * 16 * 68 <= M * L.sx && 5 * 16 * 68 <= M * L.sh;
    const int mode = L.wide ? 1 : (basis_fits ? 0 : 2);
    if (S.feature_c == 64) return nb == 1 ? pick_ntw<4, 1>(ntw, mode) : pick_ntw<4, 2>(ntw, mode);
    if (S.feature_c == 128) return nb == 1 ? pick_ntw<8, 1>(ntw, mode) : pick_ntw<8, 2>(ntw, mode);
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
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(fn), (size_t)(bytes));
    if (e != hipSuccess) return (int)e;
    if (!grads->dv_out || !grads->wslab || !grads->x_saved || !grads->h1_saved || !grads->h2_saved || !grads->rgb_fwd ||
        !shade->w1t || !shade->w2t)
        return (int)hipErrorInvalidValue;
    TileSrc src{counters, seg_cap, 0, app_ray, app_xyz, rays, ndc, nullptr, nullptr};
#ifdef TF_PHASE_TIMING
    const int n_wg = g_dbg_bwd_wgs;
#else
    const int n_wg = 256;
#endif
    const int ntw = ntw_of(kpad16(shade->in_c) / 16);
    hipLaunchKernelGGL(fn, dim3(n_wg), dim3(512), bytes, (hipStream_t)stream, *shade, src, grad_rgb, *grads);
    hipLaunchKernelGGL(wslab_reduce_kernel, dim3((unsigned)(wslab_floats(*shade) / 256)), dim3(1024), 0, (hipStream_t)stream, *shade,
                       counters, seg_cap, n_wg, ntw, *grads);
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
int tf_debug_phase_cycles_bwd_w4(unsigned long long* out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(tf_phase_cycles_w4), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(tf_phase_cycles_w4), z, sizeof(z));
    }
    return (int)e;
}
#endif

/* 1 when tf_shade_backward can differentiate this head: MLP heads of width 64 / 128, app_dim <= 32, an input of at
 * most 192 columns, and a chunk (dZ1 | V, H1, H2 | dX, X of 64 samples) that fits the CU's 160 KB of LDS. */
int tf_shade_backward_supported(const TfShade* shade) { return pick_bwd(*shade) != nullptr; }

/* floats the caller must provide in TfShadeGrads.wslab (256 workgroup slabs) */
size_t tf_shade_backward_wslab_floats(const TfShade* shade) { return 256 * wslab_floats(*shade); }

}  // extern "C"
