// tf_shade.h — pieces shared by the shading forward (shade.hip) and backward (shade_bwd.hip) kernels.
#pragma once
#include "tf_device.h"
#include "tf_sincos.h"

namespace tf {

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
    L.offInfo = L.offB + M * b;             // xyz[64][3], view[64][3]
    L.offPre = L.offInfo + M * 6;           // tile prefix over shards (65 ints) + ticket box
    L.total = L.offPre + 68;                // config 2 (144 / 150 wide): 81,680 B -> TWO workgroups per CU (160 KB)
    return L;
}

// The wave computes D[f][s] += sum_k W[f][k] * X[s][k] for feature tiles f_base+16a (a<NF) and sample
// tiles s_base+16b (b<NS).  W: global, row stride ldw (multiple of 16, zero padded); X: LDS, stride ldx.
// Lane (r = lane&15, kq = lane>>4) loads W[f][16kg+4kq..+3] and X[s][16kg+4kq..+3]; MFMA step e uses
// element e of both, i.e. k = 16kg+4kq+e on both operands.  D: row(feature) = 4*(lane>>4)+reg, col(sample) = lane&15.
// Software-pipelined in registers: the weight fragments (L2 latency, 4 VGPRs per tile) are fetched TWO k-groups
// ahead and the LDS fragments ONE ahead, so the 4*NF*NS MFMAs of a k-group issue while the next operands are in
// flight (an un-pipelined loop waits out the whole L2 round trip in front of every k-group: 59 % of the MFMA rate
// at two waves per SIMD in round 1).
template <int NF, int NS>
__device__ __forceinline__ void mma_block(const float* __restrict__ Wg, int ldw, int f_base, const float* Xs, int ldx,
                                          int s_base, int kgroups, f32x4 (&acc)[NF][NS], int lane) {
    const int r = lane & 15, kq = lane >> 4;
    const float* wp = Wg + (size_t)(f_base + r) * ldw + 4 * kq;
    const float* xp = Xs + (s_base + r) * ldx + 4 * kq;
    f32x4 a0[NF], a1[NF], b0[NS];
#pragma unroll
    for (int i = 0; i < NF; ++i) a0[i] = *reinterpret_cast<const f32x4*>(wp + (size_t)16 * i * ldw);
#pragma unroll
    for (int i = 0; i < NF; ++i)
        a1[i] = kgroups > 1 ? *reinterpret_cast<const f32x4*>(wp + (size_t)16 * i * ldw + 16) : a0[i];
#pragma unroll
    for (int j = 0; j < NS; ++j) b0[j] = *reinterpret_cast<const f32x4*>(xp + 16 * j * ldx);
#pragma unroll 1
    for (int kg = 0; kg < kgroups; ++kg) {
        f32x4 a2[NF], b1[NS];
        const int kga = kg + 2 < kgroups ? kg + 2 : kg, kgb = kg + 1 < kgroups ? kg + 1 : kg;   // tail: harmless re-reads
#pragma unroll
        for (int i = 0; i < NF; ++i) a2[i] = *reinterpret_cast<const f32x4*>(wp + (size_t)16 * i * ldw + 16 * kga);
#pragma unroll
        for (int j = 0; j < NS; ++j) b1[j] = *reinterpret_cast<const f32x4*>(xp + 16 * j * ldx + 16 * kgb);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < NS; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i][e], b0[j][e], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            a0[i] = a1[i];
            a1[i] = a2[i];
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) b0[j] = b1[j];
    }
}

// ---- the same GEMM with the weight fragments of the WHOLE phase fetched up front -------------------------------
// A phase is a handful of k-groups (<= 12): all of its weight fragments fit in registers (4 VGPRs per k-group and
// feature tile), so they are requested one phase early — before the barrier in front of the phase, behind the previous
// phase's epilogue — and the MFMA loop itself touches LDS only.  (A loop that fetches its weights as it goes leaves the
// L2 round trip in front of the first k-groups of every phase; the compiler also answers register rotation with
// s_waitcnt vmcnt(0) right behind each fetch.)
template <int NF, int KG>
__device__ __forceinline__ void load_a_frags(const float* __restrict__ Wg, int ldw, int f_base, int kgroups, int lane,
                                             f32x4 (&a)[KG][NF]) {
    const int r = lane & 15, kq = lane >> 4;
    const float* wp = Wg + (size_t)(f_base + r) * ldw + 4 * kq;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg)
#pragma unroll
        for (int i = 0; i < NF; ++i)
            // (k-groups past the end re-read the last one: unconditional loads — a guarded load is a branch and a wait of
            // its own; mma_frags never uses those fragments)
            a[kg][i] = *reinterpret_cast<const f32x4*>(wp + (size_t)16 * i * ldw + 16 * (kg < kgroups ? kg : kgroups - 1));
    // keep the requests HERE: without the fence the scheduler sinks each load down to its first use and waits for it
    // there (s_waitcnt vmcnt(0) per fragment: eight serialized L2 round trips per phase)
    __builtin_amdgcn_sched_barrier(0);
}
template <int NF, int NS, int KG>
__device__ __forceinline__ void mma_frags(const f32x4 (&a)[KG][NF], const float* Xs, int ldx, int s_base, int kgroups,
                                          f32x4 (&acc)[NF][NS], int lane) {
    // The LDS fragments are consumed in steps of PW sample tiles (PW = 2: two accumulator chains alternate, 64 cycles
    // between dependent MFMAs >= the 40-cycle latency) and fetched one step ahead: 2 x PW fragments live instead of
    // 2 x NS for a whole k-group.
    constexpr int PW = NS >= 2 ? 2 : 1, NP = NS / PW, STEPS = KG * NP;
    static_assert(NS % PW == 0, "sample tiles come in pairs");
    const int r = lane & 15, kq = lane >> 4;
    const float* xp = Xs + (s_base + r) * ldx + 4 * kq;
    f32x4 b[2][PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) b[0][j] = *reinterpret_cast<const f32x4*>(xp + 16 * j * ldx);
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        const int kg = st / NP, pr = st % NP;
        if (kg < kgroups) {
            const int kg1 = (st + 1) / NP, pr1 = (st + 1) % NP;
            if (st + 1 < STEPS && kg1 < kgroups) {
#pragma unroll
                for (int j = 0; j < PW; ++j)
                    b[(st + 1) & 1][j] = *reinterpret_cast<const f32x4*>(xp + 16 * (PW * pr1 + j) * ldx + 16 * kg1);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NF; ++i)
#pragma unroll
                    for (int j = 0; j < PW; ++j)
                        acc[i][PW * pr + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kg][i][e], b[st & 1][j][e], acc[i][PW * pr + j], 0, 0, 0);
        }
    }
}

// ---- data-gradient GEMMs of the backward on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16: 16 x the fp32 rate) -------
// An activation x is kept in LDS as ONE 32-bit word  [ hi | lo ]:  hi = the upper 16 bits of x (a bf16, truncated), lo =
// the bf16 nearest to x - hi — x to 16 mantissa bits in the 4 bytes the fp32 took.  A lane's 16-byte LDS read of four
// consecutive words IS a B fragment of the K = 32 instruction with the k index interleaved: k' = 2k holds lo_k, 2k + 1
// holds hi_k (four real k per lane, sixteen per instruction — the k = 16 kg + 4 kq + e of the fp32 path, so the same
// addresses).  The weight fragment w (fp32 from L2, split in registers) meets it twice:
//     A1 = [wh, wh] per word:  sum wh (lo + hi)        A2 = [0, wl] per word:  sum wl hi
// = wh xh + wh xl + wl xh: every product but lo.lo, error ~1e-5 of the result's largest entry (fp32 pipe: 3e-7;
// profiles/r02_mfma_bf16_split_probe.txt), two MFMAs of 16 cycles per 16 k where the fp32 path issues four of 32: 4 x.
// Only the BACKWARD's products use it (dZ1 = W2^T dZ2, dX = W1^T dZ1 and the weight-gradient GEMMs): its ReLU masks come
// from the forward's SAVED activations, so no unit can switch, and the gradient bar (2e-4 of a tensor's maximum) has room.
// The FORWARD needs fp32 accuracy — a pre-activation moved by 1e-5 switches ReLU units against the reference — and gets it
// on the same pipe from THREE pieces per operand (shade.hip: split3 / mma3).
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pack_hilo(float x) {
    const unsigned b = __float_as_uint(x), hb = b & 0xFFFF0000u;
    const unsigned r = __float_as_uint(x - __uint_as_float(hb));                 // exact
    return __uint_as_float(hb | ((r + 0x7FFFu + ((r >> 16) & 1u)) >> 16));       // lo: round to nearest even
}
// fp32 weight fragment (four consecutive k of one row) -> the two A operands above
__device__ __forceinline__ void split_weight_frag(const f32x4& w, bf16x8& a1, bf16x8& a2) {
    u32x4 p1, p2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned b = __float_as_uint(w[e]), hb = b & 0xFFFF0000u;
        const unsigned r = __float_as_uint(w[e] - __uint_as_float(hb));
        p1[e] = hb | (hb >> 16);
        p2[e] = r & 0xFFFF0000u;          // lo truncated (two VALU less per value than rounding it; the product's error
                                          // stays ~1e-5 of its largest entry: the dropped lo.lo term is the same size)
    }
    a1 = __builtin_bit_cast(bf16x8, p1);
    a2 = __builtin_bit_cast(bf16x8, p2);
}
// mma_frags with the activations as [hi | lo] words in LDS (one feature tile per wave): acc[0][j] += W X^T as above.  The
// two products of a tile go to separate accumulator chains (hi-weights, lo-weights), summed at the end, so that 2 NS
// independent MFMAs lie between dependent ones.
template <int NS, int KG>
__device__ __forceinline__ void mma_frags_hilo(const f32x4 (&a)[KG][1], const float* Xs, int ldx, int s_base, int kgroups,
                                               f32x4 (&acc)[1][NS], int lane) {
    const int r = lane & 15, kq = lane >> 4;
    const float* xp = Xs + (s_base + r) * ldx + 4 * kq;
    f32x4 lo_acc[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) lo_acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 b[2][NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) b[0][j] = *reinterpret_cast<const f32x4*>(xp + 16 * j * ldx);
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        if (kg < kgroups) {
            if (kg + 1 < KG && kg + 1 < kgroups) {
#pragma unroll
                for (int j = 0; j < NS; ++j) b[(kg + 1) & 1][j] = *reinterpret_cast<const f32x4*>(xp + 16 * j * ldx + 16 * (kg + 1));
            }
            bf16x8 a1, a2;
            split_weight_frag(a[kg][0], a1, a2);
#pragma unroll
            for (int j = 0; j < NS; ++j)
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(bf16x8, b[kg & 1][j]), acc[0][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NS; ++j)
                lo_acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, __builtin_bit_cast(bf16x8, b[kg & 1][j]), lo_acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) acc[0][j] += lo_acc[j];
}

// Products (P*m)(L*m) [VM] or (L0*L1*L2)*m [CP] of one sample, channel quads sub, sub+4, ..., written to
// vrow[0 .. n_app_total).
// `lps` lanes cooperate on one sample (lane `sub` takes channel quads sub, sub+lps, ...)
__device__ __forceinline__ void app_products(const TfShade& S, const float u[3], int sub, float* vrow, int lps = 4) {
    if (S.model == TF_MODEL_VM) {
        VmTaps t;
        make_vm_taps(S.grid, u, t);
        int coff = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int C = S.app.n_comp[i];
            const float* mk = S.app.mask[i];
            if ((C & 3) == 0 && (coff & 3) == 0) {
                for (int q = sub; q < (C >> 2); q += lps) {
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
                for (int c = sub; c < C; c += lps) {
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
            for (int q = sub; q < (C >> 2); q += lps) {
                float4_t v = lerp4(S.app.line[0], C, t0, q * 4);
                v *= lerp4(S.app.line[1], C, t1, q * 4);
                v *= lerp4(S.app.line[2], C, t2, q * 4);
                if (mk) v *= ld4(mk + q * 4);
                *reinterpret_cast<float4_t*>(vrow + q * 4) = v;
            }
        } else {
            for (int c = sub; c < C; c += lps) {
                float v = lerp1(S.app.line[0], C, t0, c) * lerp1(S.app.line[1], C, t1, c);
                v *= lerp1(S.app.line[2], C, t2, c);
                if (mk) v *= mk[c];
                vrow[c] = v;
            }
        }
    }
}

// The same products for TensorVMSplit fields read by EIGHT lanes per sample (shade_forward's gather), components a
// multiple of 4 and at most 64 per plane: a lane takes channel quads s, s + 8 of a plane and requests all (<= 12) tap
// pieces of the plane / line pair before it uses the first — the per-quad loop above waits out one round trip per quad.
// Which half of the lanes takes the second quad alternates from plane to plane (48 components = 12 quads on 8 lanes: 5 / 4
// quads per lane over the three planes instead of 6 / 3).  Returns false (nothing written) for other shapes.
__device__ __forceinline__ bool app_products_lanes8(const TfShade& S, const float u[3], int sub, float* vrow) {
    if (S.model != TF_MODEL_VM) return false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if ((S.app.n_comp[i] & 3) != 0 || S.app.n_comp[i] > 64) return false;
    VmTaps t;
    make_vm_taps(S.grid, u, t);
    int coff = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int C = S.app.n_comp[i], Q = C >> 2;
        const float* mk = S.app.mask[i];
        const int s2 = (sub + 4 * (i & 1)) & 7;
        const bool has0 = s2 < Q, has1 = s2 + 8 < Q;
        const int q[2] = {has0 ? s2 : 0, has1 ? s2 + 8 : (has0 ? s2 : 0)};      // (lanes without a quad re-read one: no branch)
        float4_t pa[2][4], la[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = q[j] * 4;
            pa[j][0] = ld4(S.app.plane[i] + (size_t)t.p[i].o00 * C + ch);
            pa[j][1] = ld4(S.app.plane[i] + (size_t)t.p[i].o01 * C + ch);
            pa[j][2] = ld4(S.app.plane[i] + (size_t)t.p[i].o10 * C + ch);
            pa[j][3] = ld4(S.app.plane[i] + (size_t)t.p[i].o11 * C + ch);
            la[j][0] = ld4(S.app.line[i] + (size_t)t.l[i].o0 * C + ch);
            la[j][1] = ld4(S.app.line[i] + (size_t)t.l[i].o1 * C + ch);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 0 ? !has0 : !has1) continue;
            const int ch = q[j] * 4;
            float4_t p, l;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                p[k] = fmaf(pa[j][3][k], t.p[i].w11, fmaf(pa[j][2][k], t.p[i].w10, fmaf(pa[j][1][k], t.p[i].w01, pa[j][0][k] * t.p[i].w00)));
                l[k] = fmaf(la[j][1][k], t.l[i].w1, la[j][0][k] * t.l[i].w0);
            }
            if (mk) {
                const float4_t m = ld4(mk + ch);
                p *= m;
                l *= m;
            }
            *reinterpret_cast<float4_t*>(vrow + coff + ch) = p * l;
        }
        coff += C;
    }
    return true;
}

// The same with FOUR lanes per sample and up to three channel quads per lane and plane (components a multiple of 4, at
// most 48 per plane): the pipelined forward's gather crew is four waves.  All (<= 18) tap pieces of a plane / line pair are
// requested before the first is used.
__device__ __forceinline__ bool app_products_lanes4(const TfShade& S, const float u[3], int sub, float* vrow) {
    if (S.model != TF_MODEL_VM) return false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if ((S.app.n_comp[i] & 3) != 0 || S.app.n_comp[i] > 48) return false;
    VmTaps t;
    make_vm_taps(S.grid, u, t);
    int coff = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int C = S.app.n_comp[i], Q = C >> 2;
        const float* mk = S.app.mask[i];
        bool has[3];
        int q[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            has[j] = sub + 4 * j < Q;
            q[j] = has[j] ? sub + 4 * j : (Q > sub ? sub : 0);      // (lanes without a quad re-read one: no branch)
        }
        float4_t pa[3][4], la[3][2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ch = q[j] * 4;
            pa[j][0] = ld4(S.app.plane[i] + (size_t)t.p[i].o00 * C + ch);
            pa[j][1] = ld4(S.app.plane[i] + (size_t)t.p[i].o01 * C + ch);
            pa[j][2] = ld4(S.app.plane[i] + (size_t)t.p[i].o10 * C + ch);
            pa[j][3] = ld4(S.app.plane[i] + (size_t)t.p[i].o11 * C + ch);
            la[j][0] = ld4(S.app.line[i] + (size_t)t.l[i].o0 * C + ch);
            la[j][1] = ld4(S.app.line[i] + (size_t)t.l[i].o1 * C + ch);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (!has[j]) continue;
            const int ch = q[j] * 4;
            float4_t p, l;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                p[k] = fmaf(pa[j][3][k], t.p[i].w11, fmaf(pa[j][2][k], t.p[i].w10, fmaf(pa[j][1][k], t.p[i].w01, pa[j][0][k] * t.p[i].w00)));
                l[k] = fmaf(la[j][1][k], t.l[i].w1, la[j][0][k] * t.l[i].w0);
            }
            if (mk) {
                const float4_t m = ld4(mk + ch);
                p *= m;
                l *= m;
            }
            *reinterpret_cast<float4_t*>(vrow + coff + ch) = p * l;
        }
        coff += C;
    }
    return true;
}

// One positional-encoding block of a 64-sample tile written into X (mlp.py:8-13): for every (sample, dim) the F
// frequencies v*2^k -> sin at x[off + d*F + k], cos at x[off + D*F + d*F + k], times the optional masks.
// `val(smp, d)` supplies v.  NT threads cooperate.  The F evaluations of an item are independent and evaluated
// branch-free two at a time unless some lane of the wave holds a huge argument (wave-uniform test).
template <int NT, typename ValFn, typename PutFn>
__device__ __forceinline__ void pe_block_put(int off, int D, int F, const float* mk, int tid, ValFn val, PutFn put, int rows = M) {
    // thread -> (sample, dim) without integer division: dims padded to a power of two
    const int dp = D <= 4 ? 4 : (D <= 32 ? 32 : 64), dsh = D <= 4 ? 2 : (D <= 32 ? 5 : 6);
    for (int it = tid; it < rows * dp; it += NT) {
        const int smp = it >> dsh, d = it & (dp - 1);
        const bool on = d < D;
        const float v = on ? val(smp, d) : 0.f;
        const float vmax = ldexpf(fabsf(v), F - 1);
        if (__builtin_expect(__any(!(vmax < 8192.f)), 0)) {        // rare: some lane needs the full range reduction
            if (on) {
                float fr = 1.f;
                for (int k = 0; k < F; ++k) {
                    float sn, cs;
                    pe_sincos(v * fr, &sn, &cs);
                    const int ci = d * F + k;
                    if (mk) {
                        sn *= mk[ci];
                        cs *= mk[D * F + ci];
                    }
                    put(smp, off + ci, sn);
                    put(smp, off + D * F + ci, cs);
                    fr *= 2.f;
                }
            }
            continue;
        }
        float fr = 1.f;
        int k = 0;
        for (; k + 1 < F; k += 2) {
            // the odd frequency of a pair by the double-angle identities: sin 2a = 2 sin a cos a, cos 2a = 1 - 2 sin^2 a
            // (|error| <= 2 x the 1.5-ulp error of the pair's first evaluation + 1 rounding: < 4e-7 absolute) — half the
            // range reductions and polynomials of the block
            float s0, c0;
            pe_sincos_fast(v * fr, &s0, &c0);
            float s1 = (2.f * s0) * c0, c1 = fmaf(-2.f * s0, s0, 1.f);
            const int ci = d * F + k;
            if (mk && on) {
                s0 *= mk[ci];
                s1 *= mk[ci + 1];
                c0 *= mk[D * F + ci];
                c1 *= mk[D * F + ci + 1];
            }
            if (on) {
                put(smp, off + ci, s0);
                put(smp, off + ci + 1, s1);
                put(smp, off + D * F + ci, c0);
                put(smp, off + D * F + ci + 1, c1);
            }
            fr *= 4.f;
        }
        if (k < F) {
            float s0, c0;
            pe_sincos_fast(v * fr, &s0, &c0);
            const int ci = d * F + k;
            if (mk && on) {
                s0 *= mk[ci];
                c0 *= mk[D * F + ci];
            }
            if (on) {
                put(smp, off + ci, s0);
                put(smp, off + D * F + ci, c0);
            }
        }
    }
}
// ... into a row-major fp32 tile X (row stride sx)
template <int NT, typename ValFn>
__device__ __forceinline__ void pe_block(float* X, int sx, int off, int D, int F, const float* mk, int tid, ValFn val,
                                         int rows = M) {
    pe_block_put<NT>(off, D, F, mk, tid, val, [&](int smp, int col, float v) { X[smp * sx + col] = v; }, rows);
}

// row / quad of item q of a [64][w4] quad image without an integer division (q < 4096, w4 <= 96: exact)
__device__ __forceinline__ void row_quad(int q, int w4, float inv_w4, int& row, int& c4) {
    row = (int)(((float)q + 0.5f) * inv_w4);
    c4 = q - row * w4;
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
    const float* view_direct;   // direct mode: (n,3) per-sample view directions instead of rays[app_ray[s]]
    const float* feat_in;       // direct mode: (n, app_dim) appearance features given by the caller (renderModule alone)
};

// Exclusive prefix of the shards' (clamped) sample counts into pre[0 .. 64] by the first 64 threads of the workgroup: one
// counter per lane and a wave scan (a single thread walking the 64 counters waits out their loads one after the other at
// the start of every workgroup).  The caller synchronises.
__device__ __forceinline__ void shard_prefix(const int* counters, int seg_cap, int* pre, int tid) {
    static_assert(TF_N_SHARDS == 64, "one shard per lane");
    if (tid < 64) {
        const int cnt = min(counters[tid * TF_SHARD_STRIDE], seg_cap);
        int inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(inc, d);
            if (tid >= d) inc += up;
        }
        pre[tid] = inc - cnt;
        if (tid == 63) pre[TF_N_SHARDS] = inc;
    }
}

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
    const int cnt = min(src.counters[lo * TF_SHARD_STRIDE], src.seg_cap);
    s0 = lo * src.seg_cap + lt * M;
    n = min(M, cnt - lt * M);
    return true;
}


}  // namespace tf
