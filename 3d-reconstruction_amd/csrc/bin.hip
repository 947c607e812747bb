// bin.hip — binned ("owner computes") scatter of the VM factor gradients.   gfx950, wave64.
//
// Why: the backward of the plane x line lookups adds 4 x C floats per (sample, plane) and 2 x C per (sample,
// line) into the gradient tensors.  Done with global float atomics this is request-bound — random 64..128-B
// pieces reach only ~0.4 TB/s on MI355X whatever the lane shape (measured) — and it was 60 % of
// the training step.  Here the samples are counting-sorted by destination first:
//     key(sample, plane i) = T x T texel tile of the footprint base     key(sample, line i) = LB-entry bucket
//   K1 count   : per-workgroup LDS histogram over the 6 keys of every entry, flushed with int atomics
//   K2 scan    : one workgroup, exclusive prefix of the histogram (+ prefix of ceil(count/chunk) work items)
//   K3 fill    : per-workgroup LDS histogram again, ONE reservation per non-empty key, LDS ranks -> binned[]
//   K4 scatter : persistent workgroups; a work item = <= chunk entries of one key; accumulate in an LDS block
//                ((T+1)^2 x C floats for a tile, (LB+1) x C for a bucket) with LDS float atomics, flush the
//                block once with contiguous global atomics (rows of (T+1) x C floats).
// Every sample's contribution is computed exactly as in the direct scatter (tf_device.h), only the
// accumulation order changes.
// Wide decompositions are handled in 16-component groups (small private blocks); the groups of a plane / line share its
// tiles / buckets, so the sort knows ONE key per (sample, plane | line) and the work-item table carries the group: a
// (key, chunk) pair appears once per group, all of them walking the same slice of binned[].  (One key per group made the
// sort's LDS atomics — 3 cycles per lane each, its whole cost — grow with the component count: 18 keys per appearance
// sample at 48 components, 54 for TensorCP at 288.)
#include "tf_device.h"

using namespace tf;

namespace {

constexpr int kCG = 16;   // components per key: wide decompositions are split into 16-component groups so that
                          // one workgroup's private accumulation blocks stay small (occupancy)

struct KeyMap {
    int T, LB;
    int tsh, lsh;             // log2(T), log2(LB): tile and bucket sizes are powers of two (no integer divides per entry)
    int ntx[3], ptiles[3], lbuckets[3], ncg[3];
    int share;                // 1: the component groups of a plane / line share its keys (the work items carry the group);
                              // 0: one key per (tile | bucket, group), laid out group-major behind plane_base / line_base
    int plane_base[3], line_base[3];
    int nkeys, keys_per_entry;
};

// cp: TensorCP has line tensors only (all with n_comp[0] components): no plane keys
__host__ __device__ inline KeyMap make_keymap(const int grid[3], const int n_comp[3], int T, int LB, bool cp, bool share) {
    KeyMap K;
    K.share = share ? 1 : 0;
    K.T = T;
    K.LB = LB;
    K.tsh = K.lsh = 0;
    while ((1 << K.tsh) < T) ++K.tsh;
    while ((1 << K.lsh) < LB) ++K.lsh;
    int run = 0, kpe = 0;
    for (int i = 0; i < 3; ++i) {
        const int W = grid[i == 2 ? 1 : 0], H = grid[i == 0 ? 1 : 2];
        K.ncg[i] = ((cp ? n_comp[0] : n_comp[i]) + kCG - 1) / kCG;
        K.ntx[i] = (W + T - 1) / T;
        K.ptiles[i] = cp ? 0 : K.ntx[i] * ((H + T - 1) / T);
        K.plane_base[i] = run;
        run += K.ptiles[i] * (share ? 1 : K.ncg[i]);
        kpe += (cp ? 1 : 2) * K.ncg[i];       // (key, group) pairs per entry: what the work-item table is sized by
    }
    for (int i = 0; i < 3; ++i) {
        K.lbuckets[i] = (grid[2 - i] + LB - 1) / LB;
        K.line_base[i] = run;
        run += K.lbuckets[i] * (share ? 1 : K.ncg[i]);
    }
    K.nkeys = run;
    K.keys_per_entry = kpe;
    return K;
}

struct SampleGeom {
    int x0[3], y0[3], l0[3];
    float fx[3], fy[3], lf[3];
};

__device__ __forceinline__ void sample_geom(const int grid[3], const float u[3], SampleGeom& g) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        tap_floor(u[mat0(i)], grid[mat0(i)], g.x0[i], g.fx[i]);
        tap_floor(u[mat1(i)], grid[mat1(i)], g.y0[i], g.fy[i]);
        tap_floor(u[vecm(i)], grid[vecm(i)], g.l0[i], g.lf[i]);
    }
}

// keys[i] (plane i) and keys[3+i] (line i)
__device__ __forceinline__ void sample_keys(const KeyMap& K, const int grid[3], const SampleGeom& g, int keys[6]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int x = min(max(g.x0[i], 0), grid[mat0(i)] - 1), y = min(max(g.y0[i], 0), grid[mat1(i)] - 1);
        const int l = min(max(g.l0[i], 0), grid[vecm(i)] - 1);
        keys[i] = K.plane_base[i] + (y >> K.tsh) * K.ntx[i] + (x >> K.tsh);
        keys[3 + i] = K.line_base[i] + (l >> K.lsh);
    }
}
#define TF_FOR_EACH_KEY(K, keys, key, BODY)                                   \
    _Pragma("unroll") for (int _i = 0; _i < 3; ++_i)                          \
        for (int _g = 0; _g < ((K).share ? 1 : (K).ncg[_i]); ++_g) {          \
            if ((K).ptiles[_i]) { const int key = (keys)[_i] + _g * (K).ptiles[_i]; BODY; } \
            { const int key = (keys)[3 + _i] + _g * (K).lbuckets[_i]; BODY; } \
        }
// component groups a key's work items fan out to (shared keys: those of the plane / line it belongs to)
__device__ __forceinline__ int key_groups(const KeyMap& K, int key) {
    if (!K.share) return 1;
    if (key >= K.line_base[0]) return K.ncg[key >= K.line_base[2] ? 2 : (key >= K.line_base[1] ? 1 : 0)];
    return K.ncg[key >= K.plane_base[2] ? 2 : (key >= K.plane_base[1] ? 1 : 0)];
}
constexpr int kItemKeyBits = 20;       // work item word 0 = key | group << 20  (TF_BIN_MAX_KEYS = 2^18 keys, <= 2^11 groups)
// The work-item table behind chunk_off[nkeys + 1]: one int4 per item {key | group << 20, first position in binned[], end
// position, 0}, written by the scan kernel.  (Round 2 stored the key word only: a scatter workgroup then chained item ->
// offsets[key], chunk_off[key] -> binned[] -> coordinates, four dependent memory round trips in front of every work item —
// 28 % of the kernel by its phase timers.  With the range in the item and the NEXT item fetched a whole item ahead the
// chain starts at binned[].)
typedef int item4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ item4_t* item_table(const TfBinJob& J, int nkeys) {
    return reinterpret_cast<item4_t*>((reinterpret_cast<uintptr_t>(J.chunk_off + nkeys + 1) + 15) & ~(uintptr_t)15);
}

// threads per workgroup and workgroups per entry shard of the count / fill passes (measured at config 2:
// 1 / 2 / 4 / 8 / 16 slices -> count 23 / 15 / 12.5 / 15 / 22 us, fill 50 / 32 / 24 / 31 / 50 us; 1024 threads: no change
// alone, and beside tf_shade_forward — where these kernels only get the ~16 CU slots that kernel leaves — the fill pass takes
// 50 us instead of 100 but the shading kernel 200 instead of 126: 0.750 ms per captured step against 0.706)
constexpr int kSortThreads = 256;
constexpr int kSlices = 4;   // workgroups per entry shard in the count / fill passes
constexpr int kSortWgs = TF_N_SHARDS * kSlices;   // workgroups per job in the count / fill passes

// The sort kernels take up to two jobs per launch (the training step sorts the density and the appearance entries
// at the same time: three launches instead of six on the second stream); blockIdx selects the job.
struct SortArgs {
    TfBinJob J[2];
    KeyMap K[2];
    int csh[2];                // log2(chunk)
};

// (a kernel rather than hipMemsetAsync: the memset issued from this library was not replayed by a captured
// hipGraph, which left the histogram un-zeroed on the second replay)
__global__ __launch_bounds__(256) void zero_ints_kernel(int* __restrict__ p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}

__device__ __forceinline__ void flag(const TfBinJob& J, int bit) {
    if (J.status) atomicOr(J.status, bit);
}

constexpr int kScanThreads = 1024;  // one 16-wave workgroup per job: the key ranges of a job are dealt over
                                   // its threads, and it reads nothing from global memory while it scans
constexpr int kKeyRange = 16384;   // keys per LDS pass of the count / scan / fill kernels (64 KB of ints)

// entries of shard g handled by slice k: local = k*256 + tid, += kSlices*256.  Jobs with more than kKeyRange keys
// (grids beyond ~400^3 at 48 components) are counted in passes over key ranges; the entries of a workgroup are
// few (cnt / kSlices), so re-deriving their keys per pass is cheap next to the LDS table work.
__global__ __launch_bounds__(kSortThreads) void bin_count_kernel(const SortArgs A) {
    extern __shared__ int lh[];
    const int which = blockIdx.x / kSortWgs, bid = blockIdx.x % kSortWgs;
    const TfBinJob& J = A.J[which];
    const KeyMap& K = A.K[which];
    const int g = bid / kSlices, k = bid % kSlices;
    const int cnt = min(J.counters[g * TF_SHARD_STRIDE + J.slot], J.seg_cap);      // (the counter holds the demand)
    for (int k0 = 0; k0 < K.nkeys; k0 += kKeyRange) {
        const int kn = min(kKeyRange, K.nkeys - k0);
        for (int i = threadIdx.x; i < kn; i += kSortThreads) lh[i] = 0;
        __syncthreads();
        for (int local = k * kSortThreads + threadIdx.x; local < cnt; local += kSlices * kSortThreads) {
            const size_t e = (size_t)g * J.seg_cap + local;
            const float u[3] = {J.xyz[e * 3], J.xyz[e * 3 + 1], J.xyz[e * 3 + 2]};
            SampleGeom sg;
            sample_geom(J.grid, u, sg);
            int keys[6];
            sample_keys(K, J.grid, sg, keys);
            TF_FOR_EACH_KEY(K, keys, key, if ((unsigned)(key - k0) < (unsigned)kn) atomicAdd(&lh[key - k0], 1));
        }
        __syncthreads();
        for (int i = threadIdx.x; i < kn; i += kSortThreads)
            if (lh[i]) atomicAdd(&J.hist[k0 + i], lh[i]);
        __syncthreads();
    }
}

// offsets[] = exclusive prefix of hist[], chunk_off[] = exclusive prefix of ceil(hist/chunk); cursor = offsets.
// One workgroup walks the keys in ranges of kKeyRange with a running carry: the range's histogram is pulled into
// LDS with coalesced loads first — every later pass touches LDS only (per-thread strided global reads made this
// kernel a chain of memory latencies).  The kernel is kept SMALL on purpose (no per-thread register copy of the
// histogram slice: 1024 threads at 128 VGPRs need a whole empty CU, and the training step runs this kernel next to
// tf_shade_forward on a second stream).
// (one workgroup per job; the job is a template parameter — indexing the by-value argument arrays with blockIdx.x made the
// compiler build pointer / key-map tables in scratch memory)
template <int JOB>
__device__ __forceinline__ void scan_job(const SortArgs& A, int* sh) {
    const TfBinJob& J = A.J[JOB];
    const KeyMap& KM = A.K[JOB];
    const int nkeys = KM.nkeys, csh = A.csh[JOB];
    const int lb0 = KM.line_base[0], lb1 = KM.line_base[1], lb2 = KM.line_base[2];
    const int pb1 = KM.plane_base[1], pb2 = KM.plane_base[2];
    const int g0 = KM.share ? KM.ncg[0] : 1, g1 = KM.share ? KM.ncg[1] : 1, g2 = KM.share ? KM.ncg[2] : 1;
#define groups_of(key) ((key) >= lb0 ? ((key) >= lb2 ? g2 : ((key) >= lb1 ? g1 : g0)) : ((key) >= pb2 ? g2 : ((key) >= pb1 ? g1 : g0)))
    constexpr int NT = kScanThreads, NWV = NT / 64;
    __shared__ int part[NWV + 1], part2[NWV + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // item table: work item -> key | group << kItemKeyBits; a key's items are laid out group by group, so the chunk index
    // is (item - chunk_off[key]) - group * chunks(key); lives behind chunk_off[]
    item4_t* items = item_table(J, nkeys);
    int* so = sh + min(nkeys, kKeyRange);       // the range's entry prefixes (offsets), kept for the item ranges
    const int cm1 = J.chunk - 1;
    int carry = 0, carry2 = 0;
    for (int k0 = 0; k0 < nkeys; k0 += kKeyRange) {
        const int kn = min(kKeyRange, nkeys - k0);
        __syncthreads();                        // the previous range's copy-out has finished reading sh / part
        for (int i = tid; i < kn; i += NT) sh[i] = J.hist[k0 + i];
        __syncthreads();
        const int kPer = (kn + NT - 1) / NT;    // keys per thread (contiguous)
        const int lo = min(kn, tid * kPer), hi = min(kn, (tid + 1) * kPer);
        int s = 0, s2 = 0;
#pragma unroll 1
        for (int i = lo; i < hi; ++i) {
            const int h = sh[i];
            s += h;
            s2 += ((h + cm1) >> csh) * groups_of(k0 + i);
        }
        // inclusive scan over the threads: shuffles inside each wave, then the wave totals
        int v = s, v2 = s2;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int a = __shfl_up(v, o, 64), b = __shfl_up(v2, o, 64);
            if (lane >= o) {
                v += a;
                v2 += b;
            }
        }
        if (lane == 63) {
            part[wv] = v;
            part2[wv] = v2;
        }
        __syncthreads();
        if (wv == 0) {      // exclusive scan of the wave totals; slot NWV = grand total
            int a = lane < NWV ? part[lane] : 0, b = lane < NWV ? part2[lane] : 0;
            const int a0 = a, b0 = b;
#pragma unroll
            for (int o = 1; o < NWV; o <<= 1) {
                const int x = __shfl_up(a, o, 64), y = __shfl_up(b, o, 64);
                if (lane >= o) {
                    a += x;
                    b += y;
                }
            }
            if (lane < NWV) {
                part[lane] = a - a0;
                part2[lane] = b - b0;
            }
            if (lane == NWV - 1) {
                part[NWV] = a;
                part2[NWV] = b;
            }
        }
        __syncthreads();
        const int wbase = part[wv], wbase2 = part2[wv], tot = part[NWV], tot2 = part2[NWV];
        // The per-thread key ranges are contiguous, so writing the prefixes straight to global memory would be one
        // cache line per lane and store; they go to LDS (in place of the histogram) and leave with coalesced stores.
        int run = carry + v + wbase - s, run2 = carry2 + v2 + wbase2 - s2;
#pragma unroll 1
        for (int i = lo; i < hi; ++i) {
            const int h = sh[i];
            sh[i] = run;
            const int nc = (h + cm1) >> csh, ng = nc ? groups_of(k0 + i) : 0;
            run += h;
            run2 += nc * ng;
        }
        __syncthreads();
        for (int i = tid; i < kn; i += NT) {
            const int o = sh[i];
            J.offsets[k0 + i] = o;
            J.cursor[k0 + i] = o;
            so[i] = o;
        }
        __syncthreads();
        // chunk prefixes: the counts are recovered from neighbouring offsets (sh holds them; the end of the thread's
        // slice is `run`)
        run2 = carry2 + v2 + wbase2 - s2;
        int prev = lo < hi ? sh[lo] : 0;
#pragma unroll 1
        for (int i = lo; i < hi; ++i) {
            const int next = i + 1 < hi ? sh[i + 1] : run;
            sh[i] = run2;
            run2 += ((next - prev + cm1) >> csh) * groups_of(k0 + i);
            prev = next;
        }
        __syncthreads();
        for (int i = tid; i < kn; i += NT) {
            const int co = sh[i];
            J.chunk_off[k0 + i] = co;
            // the key's work items, group by group.  Keys are dealt round robin: occupied tiles come in runs, and a thread
            // that owned a whole run wrote all of its items alone (the kernel's critical path).  The key's item count is
            // the difference of neighbouring prefixes (LDS): no global load in this loop — the kernel runs beside
            // tf_shade_forward, where a dependent load costs microseconds
            const int n_it = (i + 1 < kn ? sh[i + 1] : carry2 + tot2) - co;
            const int ng = n_it ? groups_of(k0 + i) : 0, nc = ng > 1 ? n_it / ng : n_it;
            const int kbeg = so[i], kend = i + 1 < kn ? so[i + 1] : carry + tot;
            for (int g = 0; g < ng; ++g)
                for (int c = 0; c < nc; ++c) {
                    const int b = kbeg + c * J.chunk;
                    if (co + g * nc + c < J.items_cap)
                        items[co + g * nc + c] = (item4_t){(k0 + i) | (g << kItemKeyBits), b, min(kend, b + J.chunk), 0};
                    else flag(J, TF_BIN_ERR_ITEMS);
                }
        }
        carry += tot;
        carry2 += tot2;
    }
    if (tid == NT - 1) {
        J.offsets[nkeys] = carry;
        J.chunk_off[nkeys] = carry2;
    }
}

#undef groups_of

__global__ __launch_bounds__(kScanThreads) void bin_scan_kernel(const SortArgs A) {
    extern __shared__ int sh[];                 // hist copy of the current key range, then its prefixes
    if (blockIdx.x == 0) scan_job<0>(A, sh);
    else scan_job<1>(A, sh);
}

__global__ __launch_bounds__(kSortThreads) void bin_fill_kernel(const SortArgs A) {
    extern __shared__ int lh[];  // per key: this workgroup's count, then its running write position in `binned`
    const int which = blockIdx.x / kSortWgs, bid = blockIdx.x % kSortWgs;
    const TfBinJob& J = A.J[which];
    const KeyMap& K = A.K[which];
    const int g = bid / kSlices, k = bid % kSlices;
    const int cnt = min(J.counters[g * TF_SHARD_STRIDE + J.slot], J.seg_cap);      // (the counter holds the demand)
    for (int k0 = 0; k0 < K.nkeys; k0 += kKeyRange) {     // key ranges, as in bin_count_kernel
        const int kn = min(kKeyRange, K.nkeys - k0);
        for (int i = threadIdx.x; i < kn; i += kSortThreads) lh[i] = 0;
        __syncthreads();
        for (int local = k * kSortThreads + threadIdx.x; local < cnt; local += kSlices * kSortThreads) {
            const size_t e = (size_t)g * J.seg_cap + local;
            const float u[3] = {J.xyz[e * 3], J.xyz[e * 3 + 1], J.xyz[e * 3 + 2]};
            SampleGeom sg;
            sample_geom(J.grid, u, sg);
            int keys[6];
            sample_keys(K, J.grid, sg, keys);
            TF_FOR_EACH_KEY(K, keys, key, if ((unsigned)(key - k0) < (unsigned)kn) atomicAdd(&lh[key - k0], 1));
        }
        __syncthreads();
        for (int i = threadIdx.x; i < kn; i += kSortThreads) {     // reserve this workgroup's range of every key it holds
            const int c = lh[i];
            if (c) lh[i] = atomicAdd(&J.cursor[k0 + i], c);
        }
        __syncthreads();
        for (int local = k * kSortThreads + threadIdx.x; local < cnt; local += kSlices * kSortThreads) {
            const size_t e = (size_t)g * J.seg_cap + local;
            const float u[3] = {J.xyz[e * 3], J.xyz[e * 3 + 1], J.xyz[e * 3 + 2]};
            SampleGeom sg;
            sample_geom(J.grid, u, sg);
            int keys[6];
            sample_keys(K, J.grid, sg, keys);
            TF_FOR_EACH_KEY(K, keys, key, if ((unsigned)(key - k0) < (unsigned)kn) {
                const int pos = atomicAdd(&lh[key - k0], 1);
                if ((unsigned)pos < (unsigned)J.binned_cap) J.binned[pos] = (int)e;
                else flag(J, TF_BIN_ERR_BINNED);
            });
        }
        __syncthreads();
    }
}

// x / d for 0 <= x < 2^20 and small d via a float reciprocal (an emulated integer divide is ~40 instructions)
__device__ __forceinline__ int fdiv(int x, float inv_d) { return (int)(((float)x + 0.5f) * inv_d); }

__device__ __forceinline__ float entry_grad(const TfBinJob& J, int e, int ch) {
    return J.grad_ld ? J.grad[(size_t)e * J.grad_ld + ch] : J.grad[e];
}

__device__ __forceinline__ int rl(int v, int k) { return __builtin_amdgcn_readlane(v, k); }
__device__ __forceinline__ float rlf(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }

// entries a wave stages per round (LDS per wave: ER x (C + 8) floats).  16: with 32 the staging rows put a workgroup at
// 33 KB of LDS, four per CU; at 27 KB and 96 VGPRs five fit (20 waves per CU to hide the stage's gathers behind):
// 0.706 -> 0.694 ms per captured step at config 2
// Lanes of a wave exchange data through the wave's own LDS rows between two phases of a round: make the writes visible
// to the other lanes AND keep the compiler from moving the following LDS reads above this point (a release-only fence
// orders the stores; the loads behind it need the acquire half).
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

__host__ __device__ inline int entries_per_round(int cmax) { (void)cmax; return 16; }
__host__ __device__ inline int lane_group(int c) { return c <= 16 ? 16 : (c <= 32 ? 32 : 64); }

// K4.  Work item = <= chunk entries of one key.  Per round each wave stages ER entries:
//   stage  (lane = entry): val[c] = dL/dprod[c] * (line | plane value)[c] * m^2 for all C components with 16-B
//          loads into LDS `pre[ER][C]`, plus the entry's block cell and tap weights in `meta[ER][8]` — every
//          global load of the round is in flight at once;
//   accum  (lanes = the footprint in memory order: row, tap, component): the wave walks its ER entries; per
//          entry one LDS read of the cell / weight / value and a plain read-add-write on the wave's PRIVATE
//          accumulation block (LDS float atomics measured ~100 cycles per wave-instruction, a plain RMW is 3 short
//          LDS ops; lanes of one instruction never collide and the LDS pipe is in order).  No global access.
//   The four private blocks are summed when the work item is flushed.
__global__ __launch_bounds__(256, 5) void bin_scatter_kernel(const TfBinJob J, const KeyMap K, int ER, int cmax) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wstride = ER * (cmax + 8);
    const int blk_floats = max((K.T + 1) * (K.T + 1), K.LB + 1) * cmax;      // one wave's accumulation block (launch sizing)
    int total = J.chunk_off[K.nkeys];
    if (total > J.items_cap) {          // never walk past the item table (a histogram that does not match the entries)
        total = J.items_cap;
        if (threadIdx.x == 0 && blockIdx.x == 0) flag(J, TF_BIN_ERR_ITEMS);
    }
    const int entry_cap = TF_N_SHARDS * J.seg_cap;
    const size_t rep = (size_t)(blockIdx.x % J.grads.n_rep) * J.grads.rep_stride;
    TF_T0();
    const item4_t* items = item_table(J, K.nkeys);
    // The NEXT work item's descriptor travels global -> LDS by DMA while the current item is processed (no registers: a
    // prefetch held in VGPRs across the item cost 50 spilled dwords at this kernel's 96-register budget): two 16-byte
    // slots behind the accumulation blocks, lanes 0..3 of wave 0 carry one int each.
    int* dslot = reinterpret_cast<int*>(smem + 4 * wstride + 4 * blk_floats);
    auto request_item = [&](int wn, int slot) {
        if (threadIdx.x < 4 && wn < total)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const int*>(items + wn) + threadIdx.x),
                                             (__attribute__((address_space(3))) void*)(dslot + 4 * slot), 4, 0, 0);
    };
    request_item((int)blockIdx.x, 0);
    int par = 0;
    for (int w = blockIdx.x; w < total; w += gridDim.x, par ^= 1) {
        TF_MARK(7);
        // per-thread coordinates from an opaque copy of the thread id: nothing derived from it is hoisted out of the
        // work-item loop, which keeps the kernel at 5 waves per SIMD (96 VGPRs; 5 dwords of it live in scratch, four of
        // them touched once per work item, one once per round)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int wave = tid >> 6, lane = tid & 63;
        float* pre = smem + wave * wstride;            // [ER][C]
        float* meta = pre + ER * cmax;                 // [ER][8]: cell (int), w00, w01, w10, w11 | cell, w0, w1
        float* blk0 = smem + 4 * wstride;              // 4 private accumulation blocks (one per wave)
        // this item's descriptor was requested one item ago (wave 0 waits for its own DMA, the barrier tells the others);
        // the next one's is requested now, into the other slot (rewritten two items later, behind two barriers)
        if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const item4_t itv = *reinterpret_cast<const item4_t*>(dslot + 4 * par);
        item4_t it;
#pragma unroll
        for (int q = 0; q < 4; ++q) it[q] = __builtin_amdgcn_readfirstlane(itv[q]);
        request_item(w + (int)gridDim.x, par ^ 1);
        const int key = it[0] & ((1 << kItemKeyBits) - 1);
        int cg = (int)((unsigned)it[0] >> kItemKeyBits);
        int beg = it[1], end = it[2];
        if ((unsigned)key >= (unsigned)K.nkeys || cg >= key_groups(K, key) || beg > end || end - beg > J.chunk) {
            if (threadIdx.x == 0) flag(J, TF_BIN_ERR_ITEMS);       // not an item: a slot the scan never wrote
            continue;
        }
        if (beg < 0 || end > J.binned_cap) {      // (ranges outside binned[] are dropped and reported)
            if (threadIdx.x == 0) flag(J, TF_BIN_ERR_BINNED);
            beg = max(beg, 0);
            end = min(end, J.binned_cap);
        }
        const bool is_line = key >= K.line_base[0];
        int i = 0, local = 0;
        if (is_line) {
            i = key >= K.line_base[2] ? 2 : (key >= K.line_base[1] ? 1 : 0);
            local = key - K.line_base[i];
            if (!K.share) {
                cg = local / K.lbuckets[i];
                local -= cg * K.lbuckets[i];
            }
        } else {
            i = key >= K.plane_base[2] ? 2 : (key >= K.plane_base[1] ? 1 : 0);
            local = key - K.plane_base[i];
            if (!K.share) {
                cg = local / K.ptiles[i];
                local -= cg * K.ptiles[i];
            }
        }
        const bool cp = J.model == TF_MODEL_CP;
        const int CF = J.factors.n_comp[cp ? 0 : i];  // components of the factor tensor (memory stride)
        const int c0 = cg * kCG, C = min(kCG, CF - c0); // this key covers components [c0, c0 + C)
        int coff = c0;                                  // column of the gradient rows (VM: the axes back to back)
        for (int q = 0; q < i && !cp; ++q) coff += J.factors.n_comp[q];
        const float* mk0 = J.factors.mask[cp ? 0 : i];
        const float* mk = mk0 ? mk0 + c0 : nullptr;
        const int ja = i == 0 ? 1 : 0, jb = i == 2 ? 1 : 2;     // CP: the other two line tensors
        const int T1 = K.T + 1;
        const int nblk = is_line ? (K.LB + 1) * C : T1 * T1 * C;
        float* blk = blk0 + wave * nblk;
        for (int q = lane; q < nblk; q += 64) blk[q] = 0.f;       // (the previous item's flush ended in a barrier)
        const int W = J.grid[mat0(i)], Hh = J.grid[mat1(i)], Gl = J.grid[vecm(i)];
        const int tyb = is_line ? 0 : (local / K.ntx[i]) * K.T, txb = is_line ? 0 : (local % K.ntx[i]) * K.T;
        const int lb0 = is_line ? local * K.LB : 0;
        const bool vec = (C & 3) == 0 && (CF & 3) == 0 && (coff & 3) == 0 && (J.grad_ld & 3) == 0;
        const int rounds = (end - beg + 4 * ER - 1) / (4 * ER);
        TF_MARK(0);
        // The entry index is fetched two rounds ahead and its coordinates one round ahead, so that a round's stage
        // depends on ONE level of random global loads (factor taps + gradient piece) instead of three in a chain
        // (index -> coordinates -> taps).
        const int LPE = 64 / ER, ent = lane / LPE, sub = lane - ent * LPE;
        // (an index is validated where it is first used, a round later: a check here would wait for the load it follows)
        auto idx_of = [&](int rd) {
            const int b = beg + (rd * 4 + wave) * ER + ent;
            return (rd < rounds && b < end) ? J.binned[b] : -1;
        };
        int e_cur = idx_of(0), e_nxt = idx_of(1);
        float u_cur[3] = {0.f, 0.f, 0.f};
        if (e_cur >= entry_cap) {       // not an entry of this list: skip it, keep the evidence
            flag(J, TF_BIN_ERR_ENTRY);
            e_cur = -1;
        }
        if (e_cur >= 0) {
            u_cur[0] = J.xyz[(size_t)e_cur * 3]; u_cur[1] = J.xyz[(size_t)e_cur * 3 + 1]; u_cur[2] = J.xyz[(size_t)e_cur * 3 + 2];
        }
        for (int rd = 0; rd < rounds; ++rd) {
            const int base = beg + (rd * 4 + wave) * ER;
            const int nk = max(0, min(ER, end - base));
            const int e_far = idx_of(rd + 2);
            float u_nxt[3] = {0.f, 0.f, 0.f};
            if (e_nxt >= entry_cap) {
                flag(J, TF_BIN_ERR_ENTRY);
                e_nxt = -1;
            }
            if (e_nxt >= 0) {
                u_nxt[0] = J.xyz[(size_t)e_nxt * 3]; u_nxt[1] = J.xyz[(size_t)e_nxt * 3 + 1]; u_nxt[2] = J.xyz[(size_t)e_nxt * 3 + 2];
            }
            // No workgroup barrier inside the rounds: the staging rows, the meta rows and the accumulation block are this
            // wave's own, and a wave's LDS operations execute in order — the four waves drift apart and fill each other's
            // memory waits instead of meeting twice per round.
            wave_sync_lds();
            TF_MARK(1);
            // ---------------- stage: LPE = 64 / ER lanes per entry, lane `sub` takes channel quads sub, sub+LPE, ...
            if (ent < nk && e_cur < 0) {     // a rejected entry contributes nothing
                if (sub == 0) {
                    float* mrow = meta + ent * 8;
                    reinterpret_cast<int*>(mrow)[0] = 0;
                    mrow[1] = mrow[2] = mrow[3] = mrow[4] = 0.f;
                }
                for (int c = sub; c < C; c += LPE) pre[ent * C + c] = 0.f;
            }
            if (ent < nk && e_cur >= 0) {
                const int e = e_cur;
                const float u[3] = {u_cur[0], u_cur[1], u_cur[2]};
                const Tap2 tp = make_tap2(u[mat0(i)], u[mat1(i)], W, Hh);
                const Tap1 tl = make_tap1(u[vecm(i)], Gl);
                float* mrow = meta + ent * 8;
                if (sub == 0) {
                    if (is_line) {
                        int l0;
                        float lf;
                        tap_floor(u[vecm(i)], Gl, l0, lf);
                        reinterpret_cast<int*>(mrow)[0] = min(max(l0, 0), Gl - 1) - lb0;
                        mrow[1] = tl.w0;             // weights are already zero for out-of-range taps
                        mrow[2] = tl.w1;
                    } else {
                        int x0, y0;
                        float fx, fy;
                        tap_floor(u[mat0(i)], W, x0, fx);
                        tap_floor(u[mat1(i)], Hh, y0, fy);
                        reinterpret_cast<int*>(mrow)[0] = (min(max(y0, 0), Hh - 1) - tyb) * T1 + min(max(x0, 0), W - 1) - txb;
                        mrow[1] = tp.w00; mrow[2] = tp.w01; mrow[3] = tp.w10; mrow[4] = tp.w11;
                    }
                }
                const float df = J.grad_ld ? 0.f : J.grad[e];
                const float* grow = J.grad_ld ? J.grad + (size_t)e * J.grad_ld + coff : nullptr;
                float* dst = pre + ent * C;
                if (vec) {
                    for (int c = 4 * sub; c < C; c += 4 * LPE) {
                        float4_t v;
                        if (cp) {     // d(L0 L1 L2 m)/dL_i = the other two lines' values (tensoRF.py:363-384)
                            v = lerp4(J.factors.line[ja], CF, make_tap1(u[vecm(ja)], J.grid[vecm(ja)]), c0 + c) *
                                lerp4(J.factors.line[jb], CF, make_tap1(u[vecm(jb)], J.grid[vecm(jb)]), c0 + c);
                        } else {
                            v = is_line ? bilerp4(J.factors.plane[i], CF, tp, c0 + c) : lerp4(J.factors.line[i], CF, tl, c0 + c);
                        }
                        v *= grow ? ld4(grow + c) : (float4_t){df, df, df, df};
                        if (mk) {
                            const float4_t m = ld4(mk + c);
                            v *= cp ? m : m * m;      // VM: each factor carries the mask; CP: the product does, once
                        }
                        *reinterpret_cast<float4_t*>(dst + c) = v;
                    }
                } else {
                    for (int c = sub; c < C; c += LPE) {
                        float v;
                        if (cp) {
                            v = lerp1(J.factors.line[ja], CF, make_tap1(u[vecm(ja)], J.grid[vecm(ja)]), c0 + c) *
                                lerp1(J.factors.line[jb], CF, make_tap1(u[vecm(jb)], J.grid[vecm(jb)]), c0 + c);
                        } else {
                            v = is_line ? bilerp1(J.factors.plane[i], CF, tp, c0 + c) : lerp1(J.factors.line[i], CF, tl, c0 + c);
                        }
                        v *= grow ? grow[c] : df;
                        if (mk) v *= cp ? mk[c] : mk[c] * mk[c];
                        dst[c] = v;
                    }
                }
            }
            wave_sync_lds();
            TF_MARK(2);
            // ---------------- accumulate (private block, plain read-add-write)
            const int nfoot = is_line ? 2 * C : 4 * C;
            for (int j = lane; j < nfoot; j += 64) {
                int row = 0, o = j;
                if (!is_line) {
                    row = j >= 2 * C;
                    o = j - (row ? 2 * C : 0);
                }
                const int tx = o >= C, c = o - (tx ? C : 0);
                const int lane_off = (row * T1 + tx) * C + c, wsel = 1 + row * 2 + tx;
                for (int k = 0; k < nk; ++k) {
                    const float* mrow = meta + k * 8;
                    const int a = reinterpret_cast<const int*>(mrow)[0] * C + lane_off;
                    blk[a] = fmaf(pre[k * C + c], mrow[wsel], blk[a]);
                }
            }
            e_cur = e_nxt;
            e_nxt = e_far;
            u_cur[0] = u_nxt[0]; u_cur[1] = u_nxt[1]; u_cur[2] = u_nxt[2];
        }
        __syncthreads();
        TF_MARK(3);
        // ---------------- flush the block: contiguous rows of (T+1) x C (or C) floats
        const float invC = 1.f / (float)C, invT1 = 1.f / (float)T1;
        if (!is_line) {
            float* gp = J.grads.plane[i];
            for (int q = tid; q < nblk; q += 256) {
                const float v = (blk0[q] + blk0[nblk + q]) + (blk0[2 * nblk + q] + blk0[3 * nblk + q]);
                if (v == 0.f) continue;
                const int cell = fdiv(q, invC), c = q - cell * C, cy = fdiv(cell, invT1);
                const int yy = tyb + cy, xx = txb + cell - cy * T1;
                if (yy < Hh && xx < W) atomicAdd(gp + ((size_t)yy * W + xx) * CF + c0 + c, v);
            }
        } else {
            float* gl = J.grads.line[i] + rep;
            for (int q = tid; q < nblk; q += 256) {
                const float v = (blk0[q] + blk0[nblk + q]) + (blk0[2 * nblk + q] + blk0[3 * nblk + q]);
                if (v == 0.f) continue;
                const int eq = fdiv(q, invC), c = q - eq * C, ent = lb0 + eq;
                if (ent < Gl) atomicAdd(gl + (size_t)ent * CF + c0 + c, v);
            }
        }
        __syncthreads();
        TF_MARK(4);
    }
    TF_FLUSH();
}

}  // namespace

extern "C" {

#ifdef TF_PHASE_TIMING
int tf_debug_phase_cycles_bin(unsigned long long* out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(tf_phase_cycles), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(tf_phase_cycles), z, sizeof(z));
    }
    return (int)e;
}
#endif

int tf_bin_status(const int* status, int* bits_out, tf_stream_t stream) {
    int bits = 0;
    if (status) {
        hipError_t e = hipMemcpyAsync(&bits, status, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    if (bits_out) *bits_out = bits;
    return bits ? (int)hipErrorAssert : 0;
}

int tf_bin_nkeys(int model, const int grid[3], const int n_comp[3], int tile, int bucket, int share) {
    return make_keymap(grid, n_comp, tile, bucket, model == TF_MODEL_CP, share != 0).nkeys;
}
int tf_bin_keys_per_entry(int model, const int n_comp[3]) {
    const int grid1[3] = {8, 8, 8};
    return make_keymap(grid1, n_comp, 8, 8, model == TF_MODEL_CP, true).keys_per_entry;
}

static bool job_shape_ok(const TfBinJob* job, KeyMap& K) {
    K = make_keymap(job->grid, job->factors.n_comp, job->tile, job->bucket, job->model == TF_MODEL_CP, job->share_groups != 0);
    if (K.nkeys != job->nkeys || K.nkeys > TF_BIN_MAX_KEYS) return false;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    return pow2(job->tile) && pow2(job->bucket) && pow2(job->chunk);
}

// count -> scan -> fill for one or two jobs (three launches either way)
static int launch_sort(const TfBinJob* const jobs[2], int n, hipStream_t st) {
    SortArgs A;
    int kr = 0;
    for (int j = 0; j < n; ++j) {
        if (!job_shape_ok(jobs[j], A.K[j])) return (int)hipErrorInvalidValue;
        A.J[j] = *jobs[j];
        A.csh[j] = 0;
        while ((1 << A.csh[j]) < jobs[j]->chunk) ++A.csh[j];
        const int r = A.K[j].nkeys < kKeyRange ? A.K[j].nkeys : kKeyRange;      // keys per LDS pass
        kr = r > kr ? r : kr;
        if (!jobs[j]->hist_zeroed)
            hipLaunchKernelGGL(zero_ints_kernel, dim3((A.K[j].nkeys + 255) / 256), dim3(256), 0, st, jobs[j]->hist, A.K[j].nkeys);
    }
    if (n == 1) {
        A.J[1] = A.J[0];
        A.K[1] = A.K[0];
        A.csh[1] = A.csh[0];
    }
    // At least 48 KB of LDS per workgroup although the shared keys need far less: in the training step these kernels run
    // beside tf_shade_forward, whose workgroups leave ~78 KB free on the CUs that hold one of them — with 18 KB histograms
    // four sort workgroups (16 waves of LDS atomics) moved in next to that one shading workgroup, and the evenly dealt
    // shading kernel waited for those CUs: 200 us instead of 150.
    const size_t lds_need = sizeof(int) * (size_t)kr, lds = lds_need < 53352 ? 53352 : lds_need;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(bin_count_kernel), (size_t)(lds));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bin_count_kernel, dim3(n * kSortWgs), dim3(kSortThreads), lds, st, A);
    const size_t lds_scan = 2 * lds_need > lds ? 2 * lds_need : lds;      // histogram / prefixes + the entry offsets of the range
    e = ensure_dynamic_lds(reinterpret_cast<const void*>(bin_scan_kernel), (size_t)(lds_scan));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bin_scan_kernel, dim3(n), dim3(kScanThreads), lds_scan, st, A);
    e = ensure_dynamic_lds(reinterpret_cast<const void*>(bin_fill_kernel), (size_t)(lds));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bin_fill_kernel, dim3(n * kSortWgs), dim3(kSortThreads), lds, st, A);
    return TF_CHECK_LAUNCH();
}

int tf_binned_sort_pair(const TfBinJob* a, const TfBinJob* b, tf_stream_t stream) {
    if (!a) return (int)hipErrorInvalidValue;
    const TfBinJob* const jobs[2] = {a, b};
    return launch_sort(jobs, b ? 2 : 1, (hipStream_t)stream);
}

int tf_binned_scatter(const TfBinJob* job, tf_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    KeyMap K;
    if (!job_shape_ok(job, K)) return (int)hipErrorInvalidValue;
    int cmax = job->factors.n_comp[0];
    for (int i = 1; i < 3 && job->model != TF_MODEL_CP; ++i) cmax = job->factors.n_comp[i] > cmax ? job->factors.n_comp[i] : cmax;
    cmax = cmax > kCG ? kCG : cmax;
    const size_t blk_bytes = (size_t)(job->tile + 1) * (job->tile + 1) * cmax * 4;
    const size_t lblk_bytes = (size_t)(job->bucket + 1) * cmax * 4;
    const int ER = entries_per_round(cmax);
    const size_t sc_bytes = 4 * (blk_bytes > lblk_bytes ? blk_bytes : lblk_bytes) + (size_t)4 * ER * (cmax + 8) * 4 + 32;   // + 2 item slots
    if (sc_bytes > 150 * 1024) return (int)hipErrorInvalidValue;
    int per_cu = (int)((160 * 1024) / (sc_bytes + 512));
    // (never more workgroups than are resident at once — 5 per CU by registers: the items are dealt by a fixed stride, and
    // workgroups that start when the first ones finish double the kernel's time)
    per_cu = per_cu > 5 ? 5 : (per_cu < 1 ? 1 : per_cu);
    if (job->stage < 0 || job->stage > 2) return (int)hipErrorInvalidValue;
    if (job->stage != 2) {
        const TfBinJob* const jobs[2] = {job, nullptr};
        const int rc = launch_sort(jobs, 1, st);
        if (rc != 0 || job->stage == 1) return rc;
    }
    if (!job->grad) return (int)hipErrorInvalidValue;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(bin_scatter_kernel), (size_t)(sc_bytes));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(256 * per_cu), dim3(256), sc_bytes, st, *job, K, ER, cmax);
    return TF_CHECK_LAUNCH();
}

}  // extern "C"
