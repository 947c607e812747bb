// tf_sincos.h — sin / cos of the positional-encoding arguments (mlp.py:8-13), fp32, registers only.
//
// |x| < 8192: Cody-Waite reduction by pi/2 in three parts + the Cephes single-precision minimax polynomials
// (<= ~1.5 ulp, ~30 VALU instructions for the pair).  Larger arguments (features times 2^k that a trained field never
// produces, but torch.sin / torch.cos define a value for): Payne-Hanek reduction on integer registers — the library
// sincosf carries a table walk through private memory (36 B of scratch per lane in round 1) and ~100 live registers,
// which every kernel that inlines it pays for.  Plain C++ so that tests/ can compile the same code on the host.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define TF_HD __host__ __device__ __forceinline__
#else
#define TF_HD static inline
#endif

namespace tf {

TF_HD void sincos_poly(float r, int q, float* sn, float* cs) {      // |r| <= pi/4, q = quadrant
    const float r2 = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f),
                          r2 * r2, fmaf(-0.5f, r2, 1.f));
    const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
    *sn = (q & 2) ? -s0 : s0;
    *cs = ((q + 1) & 2) ? -c0 : c0;
}

// |x| < 8192, branch-free
TF_HD void pe_sincos_fast(float x, float* sn, float* cs) {
    const float n = rintf(x * 0.63661977236758134f);
    float r = fmaf(-n, 1.5703125f, x);
    r = fmaf(-n, 4.837512969970703125e-4f, r);
    r = fmaf(-n, 7.54978995489188e-8f, r);
    sincos_poly(r, (int)n & 3, sn, cs);
}

// 2/pi = 0.A2F9836E 4E441529 FC2757D1 F534DDC0 DB629599 3C439041 FE5163AB DEBBC561 ... (hex), a zero word in front
#if defined(__HIPCC__)
__device__ __constant__
#endif
static const uint32_t kTwoOverPi[9] = {0u, 0xA2F9836Eu, 0x4E441529u, 0xFC2757D1u, 0xF534DDC0u, 0xDB629599u,
                                       0x3C439041u, 0xFE5163ABu, 0xDEBBC561u};

// Any finite x with |x| >= 2^13: x = m 2^e with a 24-bit integer m; frac(x / 2 pi) = frac(m * frac(2^(e-2) * 2/pi)) and
// the fraction of 2^(e-2) * 2/pi is a 96-bit window of the binary expansion of 2/pi starting e - 2 bits in.
#if defined(__HIPCC__)
__device__ __forceinline__
#else
static inline
#endif
void pe_sincos_big(float x, float* sn, float* cs) {
    uint32_t bits;
    memcpy(&bits, &x, 4);
    const uint32_t E = (bits >> 23) & 0xFFu;
    if (E == 0xFFu) {                       // inf / nan -> nan (torch.sin(inf) = nan)
        *sn = *cs = x - x;
        return;
    }
    const uint32_t m = (bits & 0x7FFFFFu) | 0x800000u;
    const uint32_t k = E - 120u;            // bit offset into {0, 2/pi bits ...}: (e - 2) + 32, e = E - 150; >= 20 here
    const uint32_t idx = k >> 5, sh = k & 31u;
    // four consecutive words of {0, 2/pi bits}; idx is 0 .. 4.  A table in constant memory read with per-lane loads: a
    // chain of selects over literals makes the compiler park all nine words in registers for the whole kernel
    const uint32_t w0 = kTwoOverPi[idx], w1 = kTwoOverPi[idx + 1], w2 = kTwoOverPi[idx + 2], w3 = kTwoOverPi[idx + 3];
    uint64_t U = ((uint64_t)w0 << 32) | w1;
    uint32_t Vw = w2;
    if (sh) {
        U = (U << sh) | (w2 >> (32u - sh));
        Vw = (w2 << sh) | (w3 >> (32u - sh));
    }
    const uint64_t p = (uint64_t)m * U + (((uint64_t)m * Vw) >> 32);      // frac(|x| / 2 pi) * 2^64 (mod 2^64)
    uint32_t q = (uint32_t)(p >> 62);
    const uint64_t t = p << 2;                                             // frac(|x| / (pi/2)) * 2^64
    q += (uint32_t)(t >> 63);                                              // nearest quadrant
    const float r = (float)((double)(int64_t)t * 8.5153039502163873e-20);  // (pi/2) / 2^64, |r| <= pi/4
    float s, c;
    sincos_poly(r, (int)(q & 3u), &s, &c);
    *sn = (bits >> 31) ? -s : s;
    *cs = c;
}

// general entry: one range test
#if defined(__HIPCC__)
__device__ __forceinline__
#else
static inline
#endif
void pe_sincos(float x, float* sn, float* cs) {
    if (fabsf(x) < 8192.f) pe_sincos_fast(x, sn, cs);
    else pe_sincos_big(x, sn, cs);
}

}  // namespace tf
